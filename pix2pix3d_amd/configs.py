"""Generator / discriminator constructor arguments of the shipped pix2pix3D configurations, as train.py of the
reference assembles them (train.py:285-316, 340-353, 374-483, 509-512).  Pure data + stdlib: importable from the
reference-side golden generator without touching this package."""
import copy

_COMMON_RENDER = dict(image_resolution=None, disparity_space_sampling=False, clamp_mode='softplus', c_gen_conditioning_zero=False,
                      gpc_reg_prob=0.5, c_scale=1.0, superresolution_noise_mode='none', density_reg=0.25, density_reg_p_dist=0.004,
                      reg_type='l1', decoder_lr_mul=1.0, sr_antialias=True)

_DATASETS = {
    # afhq cats (seg2cat) and celeba faces (seg2face): train.py:425-450
    'seg2cat': dict(res=512, sem=6, data_type='seg', sr='8XDC', nrr=128,
                    render=dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1,
                                avg_camera_radius=2.7, avg_camera_pivot=[0, 0, -0.06])),
    'seg2face': dict(res=512, sem=19, data_type='seg', sr='8XDC', nrr=128,
                     render=dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1,
                                 avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2])),
    # shapenet cars (edge2car): train.py:451-461
    'edge2car': dict(res=128, sem=1, data_type='edge', sr='2X', nrr=64,
                     render=dict(depth_resolution=64, depth_resolution_importance=64, ray_start=0.1, ray_end=2.6, box_warp=1.6, white_back=True,
                                 avg_camera_radius=1.7, avg_camera_pivot=[0, 0, 0])),
}


def dataset_info(name):
    return copy.deepcopy(_DATASETS[name])


def generator_kwargs(name, map_depth=2, cbase=32768, cmax=512, sr_num_fp16_res=4, g_num_fp16_res=0, depth=None):
    """kwargs for dnnlib.util.construct_class_by_name(...) building the generator of dataset ``name``.
    ``depth`` (S_c, S_f) overrides the sample counts (the BASELINE metric is quoted at 64+64)."""
    d = _DATASETS[name]
    rk = dict(_COMMON_RENDER, **d['render'])
    rk['image_resolution'] = d['res']
    rk['superresolution_module'] = f"training.superresolution.SuperresolutionHybrid{d['sr']}"
    rk['superresolution_module_semantic'] = f"training.superresolution.SuperresolutionHybrid{d['sr']}_semantic"
    if depth is not None:
        rk['depth_resolution'], rk['depth_resolution_importance'] = depth
    if d['data_type'] == 'seg':
        mk = dict(class_name='training.triplane_cond.MaskMappingNetwork_disentangle', in_resolution=d['res'], in_channels=d['sem'], num_layers=map_depth)
    else:
        mk = dict(class_name='training.triplane_cond.EdgeMappingNetwork_disentangle', in_resolution=d['res'], in_channels=1, num_layers=map_depth, geometry_layer=7)
    return dict(class_name='training.triplane_cond.TriPlaneSemanticEntangleGenerator', z_dim=512, w_dim=512, c_dim=25,
                img_resolution=d['res'], img_channels=3, semantic_channels=d['sem'], data_type=d['data_type'],
                mapping_kwargs=mk, rendering_kwargs=rk, channel_base=cbase, channel_max=cmax,
                fused_modconv_default='inference_only', num_fp16_res=g_num_fp16_res, conv_clamp=256 if g_num_fp16_res > 0 else None,
                sr_num_fp16_res=sr_num_fp16_res, sr_kwargs=dict(channel_base=cbase, channel_max=cmax, fused_modconv_default='inference_only'))


def oracle_cfg(name, depth=None, sr_num_fp16_res=4):
    """Matching configuration dict for oracle.model_oracle.synthesis."""
    kw = generator_kwargs(name, depth=depth, sr_num_fp16_res=sr_num_fp16_res)
    return dict(rendering_kwargs=kw['rendering_kwargs'], semantic_channels=kw['semantic_channels'], sr_kind=_DATASETS[name]['sr'],
                sr_clamp=256 if sr_num_fp16_res > 0 else None, lr_mul=kw['rendering_kwargs']['decoder_lr_mul'])


def orbit_camera(k, radius=2.7, focal=4.2647, n_frames=120, pivot=(0.0, 0.0, 0.0)):
    """25-float camera label of frame k of the orbit applications/generate_video.py:58-61, 127-137 renders:
    look-at pose (camera_utils.py:68-86, 118-137) + normalised intrinsics.  numpy only."""
    import math
    import numpy as np
    phi = 2 * math.pi * k / n_frames
    h, v = math.pi / 2 + 0.35 * math.sin(phi), math.pi / 2 - 0.05 + 0.25 * math.cos(phi)
    v = min(max(v, 1e-5), math.pi - 1e-5)
    theta, ph = h, math.acos(1 - 2 * (v / math.pi))
    pivot = np.asarray(pivot, np.float64)
    pos = pivot + radius * np.array([math.sin(ph) * math.cos(math.pi - theta), math.cos(ph), math.sin(ph) * math.sin(math.pi - theta)])
    fwd = pivot - pos
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 1.0, 0.0])
    right = -np.cross(up, fwd); right /= np.linalg.norm(right)                       # camera_utils.create_cam2world_matrix
    up2 = np.cross(fwd, right); up2 /= np.linalg.norm(up2)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up2, fwd, pos
    K = np.array([[focal, 0, 0.5], [0, focal, 0.5], [0, 0, 1]], np.float64)
    return np.concatenate([c2w.reshape(-1), K.reshape(-1)]).astype(np.float32)


def variant_kwargs(which):
    """Small instances (16-channel backbone, 128^2 output) of the generator classes train.py does not select any more: the two-backbone
    ``TriPlaneSemanticGenerator``, ``..._withBG``, and the image-only ``TriPlaneGenerator`` over the entangled mapping networks.
    Used by the golden generator and the tests of those classes."""
    kw = generator_kwargs('edge2car', cbase=1024, cmax=16)
    seg = dict(class_name='training.triplane_cond.MaskMappingNetwork_disentangle', in_resolution=128, in_channels=6, num_layers=2)
    if which == 'two_backbone':          # label maps, texture + semantic plane sets
        kw.update(class_name='training.triplane_cond.TriPlaneSemanticGenerator', semantic_channels=6, data_type='seg',
                  mapping_kwargs=dict(seg, class_name='training.triplane_cond.MaskMappingNetwork'))   # z_dim = 0 needs the entangled network
    elif which == 'with_bg':             # label maps, background panorama
        kw.update(class_name='training.triplane_cond.TriPlaneSemanticEntangleGenerator_withBG', semantic_channels=6, data_type='seg', mapping_kwargs=seg)
    elif which == 'with_bg_edge':        # edge maps: the background's label half is not pinned to class 0
        kw.update(class_name='training.triplane_cond.TriPlaneSemanticEntangleGenerator_withBG')
    elif which in ('mask_entangled', 'edge_entangled'):
        for k in ('semantic_channels', 'data_type'):
            kw.pop(k)
        kw['class_name'] = 'training.triplane_cond.TriPlaneGenerator'
        if which == 'mask_entangled':
            kw['mapping_kwargs'] = dict(seg, class_name='training.triplane_cond.MaskMappingNetwork')
        else:
            kw['mapping_kwargs'] = dict(kw['mapping_kwargs'], class_name='training.triplane_cond.EdgeMappingNetwork')
    else:
        raise KeyError(which)
    return kw


VARIANTS = ('two_backbone', 'with_bg', 'with_bg_edge', 'mask_entangled', 'edge_entangled')


def small_train_kwargs():
    """A training triple small enough for the CPU (128^2 label maps, 64^2 rays x 12+12 samples, 16-channel backbone, 2X SR heads): the
    generator class, label-map mapping network and discriminators that train_scripts/afhq_seg.sh selects (train.py:374-380, 509-512,
    528-531; training_loop.py:299-308), scaled down.  Returns (G kwargs, D kwargs, D_semantic kwargs).  Used by the golden recorder of
    the training phases (reference modules) and by the tests that replay them (this package's modules)."""
    g = generator_kwargs('edge2car', cbase=1024, cmax=16, depth=(12, 12))
    g.update(semantic_channels=6, data_type='seg',
             mapping_kwargs=dict(class_name='training.triplane_cond.MaskMappingNetwork_disentangle', in_resolution=128, in_channels=6, num_layers=2))
    d = dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=128, img_channels=3, channel_base=1024, channel_max=32,
             num_fp16_res=0, conv_clamp=None, disc_c_noise=0, block_kwargs=dict(freeze_layers=0), mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=2))
    ds = dict(copy.deepcopy(d), img_channels=3 + 6)                   # training_loop.py:308: image + label channels
    return g, d, ds


# Pix2Pix3DLoss arguments of train_scripts/afhq_seg.sh (train.py:331-341, 469-483, 496-501: --resume disables the blur and the swapping
# ramp), with the LPIPS weight as the script sets it
AFHQ_SEG_LOSS = dict(r1_gamma=5, blur_init_sigma=0, blur_fade_kimg=200.0, gpc_reg_prob=0.5, gpc_reg_fade_kimg=0, dual_discrimination=True,
                     neural_rendering_resolution_initial=128, neural_rendering_resolution_final=None, neural_rendering_resolution_fade_kimg=1000,
                     filter_mode='antialiased', style_mixing_prob=0, random_c_prob=0.5, lambda_l1=0, lambda_lpips=1, lambda_D_semantic=0.1,
                     seg_weight=0, edge_weight=2, only_raw_recons=True, silhouette_loss=False, lambda_cross_view=1e-4)
