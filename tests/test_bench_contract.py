"""The benchmark line's contract, checked on the line this round's tree printed on an MI355X (profiles/round6_bench_line_default.json) and on
bench.py's argument surface: the keys the driver parses, the roofline object backed by committed counter passes whose hash matches the
kernel sources of this tree, the CPU baseline, the exact-fp32 leg and the six-phase training step."""
import hashlib
import json
import os
import subprocess
import sys

from conftest import ROOT


def _line(name):
    return json.load(open(os.path.join(ROOT, 'profiles', name)))


def test_default_line_has_the_contract_keys():
    d = _line('round6_bench_line_default.json')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline', 'cpu_baseline', 'exact_fp32', 'train_step'):
        assert k in d, k
    assert d['unit'] == 'img/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 4 * d['n_gpus'] / (d['ms_per_step'] * 1e-3)) < 0.01 * d['value']           # value = images of all ranks / time
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'binding', 'tap_bytes', 'ms_per_launch', 'units_per_launch'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0 < r['frac'] <= 1.0                  # a bound: never above its peak
    assert r['binding'] is not None and r['bound'] in r['binding']['all_resources_in_pmc_pass']
    assert r['traffic'] < r['units_per_launch'] * r['tap_bytes']['bytes_per_unit']                       # memory-side bytes << tap bytes
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'img/s'
    e = d['exact_fp32']
    assert e['value'] < d['value'] and e['mfma_conv']['conv_f32']['tflops'] is not None and e['roofline']['bound'] == 'mfma_pipe'
    t = d['train_step']
    assert set(t['phase_ms']) == {'Gmain', 'Greg', 'Dmain', 'Dreg', 'D_semanticmain', 'D_semanticreg', 'ema'}          # config 3 with --dis_mask=True: six phases
    assert t['lazy_schedule']['ms_per_iteration'] < t['ms_per_iteration'] and abs(sum(t['phase_ms'].values()) - t['ms_per_iteration']) < 0.05 * t['ms_per_iteration']
    assert t['phase_ms']['Gmain'] > 2.5 * t['phase_ms']['Dmain']                       # four generator passes, two of them differentiated
    g = t['generator_bf16x3']
    assert g['ms_per_iteration'] < t['ms_per_iteration'] and abs(g['phase_ms']['Dreg'] - t['phase_ms']['Dreg']) < 0.1 * t['phase_ms']['Dreg']     # the discriminators' arithmetic is untouched
    r = t['roofline']
    assert r is None or (r['dominant']['bound'] in ('mfma', 'valu_issue', 'lds', 'hbm') and 0 < r['dominant']['frac'] <= 1)
    # round 5: the other BASELINE configurations, the real-data MFMA ceiling, the matrix-pipe floor of every training phase
    assert 'rccl' in d and d['rccl'] is None                                            # one GPU: no collective library in the line
    cfgs = d['configs']
    assert len(cfgs) == 3 and any('edge2car' in k for k in cfgs) and any('seg2face' in k for k in cfgs) and any('48+48' in k for k in cfgs)
    for k, v in cfgs.items():
        assert v['value'] > 0 and v['launch'] == 'hipgraph' and v['ray_marcher']['ms_per_launch'] > 0 and set(v['stage_ms']) == {'backbone', 'render', 'sr'}, k
        assert abs(v['ray_marcher']['ray_samples_per_s'] - v['ray_marcher']['ray_samples_per_launch'] / (v['ray_marcher']['ms_per_launch'] * 1e-3)) < 1e-3 * v['ray_marcher']['ray_samples_per_s']
    c = d['mfma_real_data_ceiling']
    assert 0.6 < c['frac_of_2p5pf'] < 0.75 and c['zero_operands_tflops'] > 2300 and 1.5 < c['clock_ghz'] < 1.9
    assert abs(c['frac_of_it']['conv_f16'] - d['mfma_conv']['conv_f16']['tflops'] / c['tflops']) < 2e-3 and c['frac_of_it']['conv_f16'] < 1
    fl = t['arithmetic_floor']['phases']
    assert set(fl) == {'Gmain', 'Greg', 'Dmain', 'Dreg', 'D_semanticmain', 'D_semanticreg'}
    for k, v in fl.items():
        assert 0 < v['floor_ms_at_quoted_peaks'] < v['measured_ms'] and v['measured_over_floor'] > 1, k                 # a floor
    f32 = lambda ph: fl[ph]['tflop']['f32_convs'] + fl[ph]['tflop'].get('bf16x6_convs_fp32_equivalent', 0.0)      # fp32 tensors, whichever pipe forms the products
    assert f32('Gmain') > 3 * f32('Dmain') and fl['Gmain']['tflop']['f32_decoder_mlps'] > 0
    # round 5, bf16x6 (fp32-accurate products on the bf16 pipe, the default): the exact leg carries the same loop with the backbone's products formed that way,
    # the training iteration the variant with every product back on the f32-input MFMA
    x6 = e['backbone_as_bf16x6']
    assert x6['value'] > e['value'] and x6['mfma_conv']['conv_bf16x6']['tflops'] > e['mfma_conv']['conv_f32']['tflops']
    assert fl['Gmain']['tflop']['bf16x6_convs_fp32_equivalent'] > fl['Gmain']['tflop']['f32_convs']
    assert t['f32_input_mfma']['ms_per_iteration'] > t['ms_per_iteration']
    # round 6: the iteration's optimizer form is named in the line
    assert 'fused=True' in t['optimizer']


def test_committed_counter_passes_belong_to_this_trees_kernel():
    h = hashlib.sha256()
    for rel in ('pix2pix3d_amd/csrc/render.hip', 'pix2pix3d_amd/csrc/render_device.h'):
        h.update(open(os.path.join(ROOT, rel), 'rb').read())
    for name in ('render_pmc.json', 'render_pmc_exact_fp32.json'):
        pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))
        assert pmc['kernel_src_sha16'] == h.hexdigest()[:16], f'profiles/{name} was taken from other kernel sources: re-run tests/gpu_pmc_render.py'
        b = pmc['binding']
        assert b['resource'] == max(pmc['derived']['utilisation'].items(), key=lambda kv: kv[1])[0] and b['busy_units_per_launch'] > 0


def test_bench_argument_surface():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ('--gpus', '--steps', '--warmup', '--train-step', '--no-exact-fp32', '--no-cpu-baseline', '--cpu-reps', '--no-configs'):
        assert flag in r.stdout, flag


def test_multi_gpu_line_refuses_the_reference_scripts_p2p_switch(monkeypatch):
    """train_scripts/afhq_seg.sh:2 exports NCCL_P2P_DISABLE=1; with it RCCL leaves xGMI.  bench.collective_info (the `rccl` object of every N > 1
    line) refuses to measure in that state — before touching a device or the process group."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv('NCCL_P2P_DISABLE', '1')
    with pytest.raises(AssertionError, match='NCCL_P2P_DISABLE'):
        bench.collective_info(None, 2, 0, None)
