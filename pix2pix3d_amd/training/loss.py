"""The training phases that drive the hot path: ``Pix2Pix3DLoss.accumulate_gradients`` restated.

Reference: training/loss.py:372-1003 (class ``Pix2Pix3DLoss``; line numbers below are that file's).  The reference's own loss.py runs unchanged on
this package through ``dropin.install(reference_root=...)`` (tests/test_loss_phases.py drives it that way); this module exists because bench.py's
``--train-step`` workload and the ``-m gpu`` parity tests need the same phases WITHOUT a reference checkout, which the GPU box does not have.  Same
constructor arguments, same ``accumulate_gradients(phase, batch, gen_z, gen_c, gain, cur_nimg)`` contract, same draws from the global generator in the
same order — so both are pinned by one set of gradients recorded from the reference (tests/golden/make_golden.py, group ``loss_phases``).

What a phase costs on the device (config 3, per GPU):
  Gmain            G.mapping + G.synthesis under autograd, D (+ D_semantic) on the result, backward; then the cross-view block (:657-678): a no-grad
                   render from gen_c, its arg-max label map pushed through G.mapping + G.synthesis under autograd, a no-grad reconstruction, backward
                   of the smooth-L1 between the two label renderings — FOUR generator passes, two of them differentiated;
  Greg             density regularisation on G.sample_mixed (:681-706 and the two 'monotonic' variants :708-824);
  Dmain / D_semanticmain    one no-grad generator pass, the discriminator on generated and on real input, two backwards (:827-869, :895-975);
  Dreg / D_semanticreg      R1 on real input: gradient of the logits w.r.t. both images under ``no_weight_gradients``, double backward (:871-893, :977-1000).

Two host-side pieces are injectable because their reference versions need things this image does not have: ``lpips`` (a callable (a, b) -> [N,1,1,1];
the reference builds ``lpips.LPIPS(net='vgg')``, a pretrained VGG: with ``lambda_lpips > 0`` and no callable this module tries that import and says so if
it fails) and ``report`` (a callable (name, value); the reference logs through torch_utils.training_stats).
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..torch_utils.ops import conv2d_gradfix, upfirdn2d
from .dual_discriminator import filtered_resizing

PHASES = ('Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth', 'D_semanticmain', 'D_semanticreg', 'D_semanticboth')

# class weights of the 19 CelebAMask labels (:409-422): --seg_weight 1 / 2
_SEG_WEIGHTS = {
    1: [0.42768099, 0.45614868, 1.59952169, 4.38863045, 4.85695198, 4.86439145, 3.53563349, 3.57896961, 3.37838867, 3.66981824, 4.17743386, 3.5624441,
        2.78190484, 0.40917425, 2.38560636, 4.65813434, 17.17367367, 1.13303585, 1.25281865],
    2: [1.82911031e-01, 2.08071618e-01, 2.55846962e+00, 1.92600773e+01, 2.35899825e+01, 2.36623042e+01, 1.25007042e+01, 1.28090235e+01, 1.14135100e+01,
        1.34675659e+01, 1.74509537e+01, 1.26910080e+01, 7.73899453e+00, 1.67423571e-01, 5.69111768e+00, 2.16982155e+01, 2.94935067e+02, 1.28377023e+00,
        1.56955458e+00],
}


def cross_entropy2d(logits, target, weight=None):
    """training/loss_utils.py:4-19: mean (class-weighted) cross entropy of [N,C,H,W] logits against an [N,Ht,Wt] index map; logits are resized
    (bilinear, align_corners) when the map has another size."""
    if tuple(logits.shape[2:]) != tuple(target.shape[1:]):
        logits = F.interpolate(logits, size=tuple(target.shape[1:]), mode='bilinear', align_corners=True)
    return F.cross_entropy(logits.permute(0, 2, 3, 1).reshape(-1, logits.shape[1]), target.reshape(-1), weight=weight, reduction='mean')


def _gaussian_blur(img, sigma):
    """The discriminator-input blur of the first kimgs (:459-464): separable 2^(-(x/sigma)^2) taps out to 3 sigma, normalised."""
    radius = np.floor(sigma * 3)
    if radius <= 0:
        return img
    taps = torch.arange(-radius, radius + 1, device=img.device).div(sigma).square().neg().exp2()
    return upfirdn2d.filter2d(img, taps / taps.sum())


class Loss:
    def accumulate_gradients(self, phase, batch, gen_z, gen_c, gain, cur_nimg):
        raise NotImplementedError


class Pix2Pix3DLoss(Loss):
    def __init__(self, device, G, D, D_semantic=None, augment_pipe=None, r1_gamma=10, style_mixing_prob=0, pl_weight=0, pl_batch_shrink=2, pl_decay=0.01,
                 pl_no_weight_grad=False, blur_init_sigma=0, blur_fade_kimg=0, r1_gamma_init=0, r1_gamma_fade_kimg=0, neural_rendering_resolution_initial=64,
                 neural_rendering_resolution_final=None, neural_rendering_resolution_fade_kimg=0, gpc_reg_fade_kimg=1000, gpc_reg_prob=None,
                 dual_discrimination=False, filter_mode='antialiased', random_c_prob=0, lambda_l1=2, lambda_lpips=10, lambda_D_semantic=1, seg_weight=0,
                 edge_weight=2, only_raw_recons=False, silhouette_loss=False, lambda_cross_view=0, lpips=None, report=None):
        assert gpc_reg_prob is None or 0 <= gpc_reg_prob <= 1
        self.device, self.G, self.D, self.D_semantic, self.augment_pipe = device, G, D, D_semantic, augment_pipe
        self.r1_gamma, self.style_mixing_prob = r1_gamma, style_mixing_prob
        self.pl_weight, self.pl_batch_shrink, self.pl_decay, self.pl_no_weight_grad = pl_weight, pl_batch_shrink, pl_decay, pl_no_weight_grad     # (kept, unused: :383-387)
        self.pl_mean = torch.zeros([], device=device)
        self.blur_init_sigma, self.blur_fade_kimg = blur_init_sigma, blur_fade_kimg
        self.r1_gamma_init, self.r1_gamma_fade_kimg = r1_gamma_init, r1_gamma_fade_kimg
        self.neural_rendering_resolution_initial = neural_rendering_resolution_initial
        self.neural_rendering_resolution_final = neural_rendering_resolution_final
        self.neural_rendering_resolution_fade_kimg = neural_rendering_resolution_fade_kimg
        self.gpc_reg_fade_kimg, self.gpc_reg_prob = gpc_reg_fade_kimg, gpc_reg_prob
        self.dual_discrimination, self.filter_mode = dual_discrimination, filter_mode
        self.resample_filter = upfirdn2d.setup_filter([1, 3, 3, 1], device=device)
        self.blur_raw_target = True
        self.random_c_prob = random_c_prob
        self.lambda_l1, self.lambda_lpips, self.lambda_D_semantic = lambda_l1, lambda_lpips, lambda_D_semantic
        self.seg_weight = torch.tensor(_SEG_WEIGHTS[int(seg_weight)]).to(device) if int(seg_weight) in _SEG_WEIGHTS else None
        self.edge_weight, self.only_raw_recons, self.silhouette_loss, self.lambda_cross_view = edge_weight, only_raw_recons, silhouette_loss, lambda_cross_view
        if lpips is None and lambda_lpips != 0:
            try:
                import lpips as _lpips_pkg                                   # :406: lpips.LPIPS(net='vgg')
                lpips = _lpips_pkg.LPIPS(net='vgg').to(device=device)
            except ImportError as e:
                raise ImportError('Pix2Pix3DLoss: lambda_lpips > 0 needs the `lpips` package (a pretrained VGG) or an `lpips=` callable (a, b) -> [N,1,1,1]; '
                                  'pass lambda_lpips=0 to train without the perceptual term') from e
        self.lpips_loss = lpips
        self.report = report if report is not None else (lambda name, value: None)

    # ------------------------------------------------------------------------------------------------------------ generator / discriminators
    def run_G(self, z, c, batch, swapping_prob, neural_rendering_resolution, update_emas=False, mode='random_z_image_c'):
        """:433-455.  ws always comes from the label map and the IMAGE's pose; 'random_z_random_c' renders it from ``c`` instead."""
        assert mode in ('random_z_image_c', 'random_z_random_c')
        ws = self.G.mapping(z, batch['pose'], batch, update_emas=update_emas)
        view = batch['pose'] if mode == 'random_z_image_c' else c
        return self.G.synthesis(ws, view, neural_rendering_resolution=neural_rendering_resolution, update_emas=update_emas), ws

    def _discriminate(self, net, img, c, blur_sigma, update_emas=False):
        pair = {'image': _gaussian_blur(img['image'].clone(), blur_sigma), 'image_raw': img['image_raw'].clone()}
        if self.augment_pipe is not None:                                # :466-473: both images through ONE augmentation, the raw one at full size
            n_ch, full, small = pair['image'].shape[1], pair['image'].shape[2:], pair['image_raw'].shape[2:]
            both = self.augment_pipe(torch.cat([pair['image'], F.interpolate(pair['image_raw'], size=full, mode='bilinear', antialias=True)], dim=1))
            pair = {'image': both[:, :n_ch], 'image_raw': F.interpolate(both[:, n_ch:], size=small, mode='bilinear', antialias=True)}
        return net(pair, c, update_emas=update_emas)

    def run_D(self, img, c, blur_sigma=0, blur_sigma_raw=0, update_emas=False):
        return self._discriminate(self.D, img, c, blur_sigma, update_emas)                       # :457-478

    def run_D_semantic(self, img, c, blur_sigma=0, blur_sigma_raw=0, update_emas=False):
        return self._discriminate(self.D_semantic, img, c, blur_sigma, update_emas)              # :480-507

    def _with_labels(self, gen_img, detach_rgb):
        """Input of D_semantic for a generated sample: image channels followed by the label channels (soft-max for label maps), at both resolutions
        (:573-590 with the rgb half detached — the generator is steered through the labels only — and :907-921 as is)."""
        out = {}
        for key, sem in (('image', 'semantic'), ('image_raw', 'semantic_raw')):
            labels = torch.softmax(gen_img[sem], dim=1) if self.G.data_type == 'seg' else gen_img[sem]
            out[key] = torch.cat([gen_img[key].detach() if detach_rgb else gen_img[key], labels], dim=1)
        return out

    def _r1(self, logits, inputs):
        """R1 penalty per image: squared gradient of the logits w.r.t. the input images, differentiable (:871-887)."""
        wanted = list(inputs) if self.dual_discrimination else list(inputs)[:1]
        with conv2d_gradfix.no_weight_gradients():
            grads = torch.autograd.grad(outputs=[logits.sum()], inputs=wanted, create_graph=True, only_inputs=True)
        return sum(g.square().sum([1, 2, 3]) for g in grads)

    # ------------------------------------------------------------------------------------------------------------ schedule
    def _schedule(self, cur_nimg):
        blur_sigma = max(1 - cur_nimg / (self.blur_fade_kimg * 1e3), 0) * self.blur_init_sigma if self.blur_fade_kimg > 0 else 0
        ramp = min(cur_nimg / (self.gpc_reg_fade_kimg * 1e3), 1) if self.gpc_reg_fade_kimg > 0 else 1
        swapping_prob = (1 - ramp) * 1 + ramp * self.gpc_reg_prob if self.gpc_reg_prob is not None else None
        nrr = self.neural_rendering_resolution_initial
        if self.neural_rendering_resolution_final is not None:
            a = min(cur_nimg / (self.neural_rendering_resolution_fade_kimg * 1e3), 1)
            nrr = int(np.rint(self.neural_rendering_resolution_initial * (1 - a) + self.neural_rendering_resolution_final * a))
        return blur_sigma, swapping_prob, nrr

    def _pick_view(self, batch, gen_c):
        """:526-531 (and again at the top of Dmain / D_semanticmain): with probability random_c_prob the sample is rendered from gen_c."""
        if torch.rand(1) < self.random_c_prob:
            return 'random_z_random_c', gen_c
        return 'random_z_image_c', batch['pose']

    # ------------------------------------------------------------------------------------------------------------ the phases
    def accumulate_gradients(self, phase, batch, gen_z, gen_c, gain, cur_nimg, debug=False):
        assert phase in PHASES
        if self.G.rendering_kwargs.get('density_reg', 0) == 0:
            phase = {'Greg': 'none', 'Gboth': 'Gmain'}.get(phase, phase)
        if self.r1_gamma == 0:
            phase = {'Dreg': 'none', 'Dboth': 'Dmain'}.get(phase, phase)
        blur_sigma, swapping_prob, nrr = self._schedule(cur_nimg)
        mode, c_render = self._pick_view(batch, gen_c)
        real_raw = filtered_resizing(batch['image'], size=nrr, f=self.resample_filter, filter_mode=self.filter_mode)
        if self.blur_raw_target:
            real_raw = _gaussian_blur(real_raw, blur_sigma)
        real = {'image': batch['image'], 'image_raw': real_raw}
        ctx = dict(batch=batch, gen_z=gen_z, gen_c=gen_c, gain=gain, blur_sigma=blur_sigma, swapping_prob=swapping_prob, nrr=nrr, real=real)
        if phase in ('Gmain', 'Gboth'):
            self._gmain(ctx, mode, c_render)
        if phase in ('Greg', 'Gboth') and self.G.rendering_kwargs.get('density_reg', 0) > 0:
            self._greg(ctx)
        if phase in ('Dmain', 'Dreg', 'Dboth'):
            self._d_phase(ctx, phase, semantic=False)
        if phase in ('D_semanticmain', 'D_semanticreg', 'D_semanticboth'):
            self._d_phase(ctx, {'D_semanticmain': 'Dmain', 'D_semanticreg': 'Dreg', 'D_semanticboth': 'Dboth'}[phase], semantic=True)

    def _render(self, ctx, batch=None, mode='random_z_image_c', update_emas=False):
        return self.run_G(ctx['gen_z'], ctx['gen_c'], ctx['batch'] if batch is None else batch, swapping_prob=ctx['swapping_prob'],
                          neural_rendering_resolution=ctx['nrr'], update_emas=update_emas, mode=mode)

    def _gmain(self, ctx, mode, c_render):
        batch, real, nrr, seg = ctx['batch'], ctx['real'], ctx['nrr'], self.G.data_type == 'seg'
        gen, _ = self._render(ctx, mode=mode)
        logits = self.run_D(gen, c_render, blur_sigma=ctx['blur_sigma'])
        self.report('Loss/scores/fake', logits); self.report('Loss/signs/fake', logits.sign())
        loss = F.softplus(-logits)                                                      # [N, 1]
        if self.D_semantic is not None:
            logits_sem = self.run_D_semantic(self._with_labels(gen, detach_rgb=True), c_render, blur_sigma=ctx['blur_sigma'])
            self.report('Loss/scores/fake_semantic', logits_sem); self.report('Loss/signs/fake_semantic', logits_sem.sign())
            loss = loss + F.softplus(-logits_sem) * self.lambda_D_semantic
        recon = sem_recon = silhouette = 0
        if mode == 'random_z_image_c':                                                  # the sample should reproduce the training image and its label map
            full_weight = 1 - float(self.only_raw_recons)

            def image_term(key):
                t = F.smooth_l1_loss(gen[key], real[key]) * self.lambda_l1
                return t + self.lpips_loss(gen[key], real[key]) * self.lambda_lpips if self.lpips_loss is not None else t.reshape(1, 1, 1, 1)
            recon = image_term('image') * full_weight + image_term('image_raw')          # [N or 1, 1, 1, 1]
            loss = loss + recon.squeeze(-1).squeeze(-1)
            if 'semantic' in gen:
                mask_raw = F.interpolate(batch['mask'], size=nrr, mode='nearest')
                if seg:
                    sem_recon = (cross_entropy2d(gen['semantic'], batch['mask'].squeeze(1).long(), weight=self.seg_weight) * full_weight
                                 + cross_entropy2d(gen['semantic_raw'], mask_raw.squeeze(1).long(), weight=self.seg_weight))
                else:
                    sem_recon = (F.smooth_l1_loss(gen['semantic'], batch['mask']) * self.edge_weight * full_weight
                                 + F.smooth_l1_loss(gen['semantic_raw'], mask_raw) * self.edge_weight)
                loss = loss + sem_recon.squeeze(-1).squeeze(-1)
                if self.silhouette_loss and seg:
                    silhouette = self.calculate_silhouette_loss(gen['weight'], mask_raw.long())
                    loss = loss + silhouette
        self.report('Loss/G/loss_img_reconstruction', recon); self.report('Loss/G/loss_semantic_reconstruction', sem_recon)
        self.report('Loss/G/loss_silhouette', silhouette); self.report('Loss/G/loss', loss)
        loss.mean().mul(ctx['gain']).backward()

        # cross-view consistency (:657-678): what the sample's label map looks like from gen_c, fed back as the conditioning input, must render
        # (from the image's pose) the label map the original conditioning renders
        with torch.no_grad():
            other_view, _ = self._render(ctx, mode='random_z_random_c')
        projected = dict(batch)
        projected['mask'] = torch.argmax(other_view['semantic'].detach(), dim=1, keepdim=True) if seg else other_view['semantic'].detach()
        from_projection, _ = self._render(ctx, batch=projected)
        with torch.no_grad():
            reconstruction, _ = self._render(ctx)
        loss_cv = F.smooth_l1_loss(from_projection['semantic_raw'], reconstruction['semantic_raw']) * self.lambda_cross_view
        self.report('Loss/G/loss_cross_view', loss_cv)
        loss_cv.mean().mul(ctx['gain']).backward()

    def _density_pair(self, ctx, n_points, perturb):
        """sigma at ``n_points`` uniform points per image and at their perturbed copies, through G.sample_mixed; the unused pose-swap draw of the
        reference (:683-687) is made too, so the generator state advances identically."""
        gen_c = ctx['gen_c']
        if ctx['swapping_prob'] is not None:
            _ = torch.where(torch.rand([], device=gen_c.device) < ctx['swapping_prob'], torch.roll(gen_c.clone(), 1, 0), gen_c)
        ws = self.G.mapping(ctx['gen_z'], ctx['batch']['pose'], ctx['batch'], update_emas=False)
        initial = torch.rand((ws.shape[0], n_points, 3), device=ws.device) * 2 - 1
        coords = torch.cat([initial, perturb(initial)], dim=1)
        sigma = self.G.sample_mixed(coords, torch.randn_like(coords), ws, update_emas=False)['sigma']
        half = sigma.shape[1] // 2
        return sigma[:, :half], sigma[:, half:]

    def _greg(self, ctx):
        rk, gain = self.G.rendering_kwargs, ctx['gain']

        def tv(dist):
            a, b = self._density_pair(ctx, 1000, lambda p: p + torch.randn_like(p) * dist)
            (F.l1_loss(a, b) * rk['density_reg']).mul(gain).backward()
        if rk['reg_type'] == 'l1':                                                       # :681-706
            tv(rk['density_reg_p_dist'])
        elif rk['reg_type'] in ('monotonic-detach', 'monotonic-fixed'):                   # :708-824: density must not drop going away from the camera, then the TV term
            step = (1 / 256) * rk['box_warp']
            front, behind = self._density_pair(ctx, 2000, lambda p: p + torch.tensor([0, 0, -1], device=p.device) * step)
            if rk['reg_type'] == 'monotonic-detach':
                front = front.detach()
            (torch.relu(front - behind).mean() * 10).mul(gain).backward()
            tv(step)

    def _d_phase(self, ctx, phase, semantic):
        """Dmain / Dreg / Dboth for D (:827-893) or, with the label channels appended to every input, for D_semantic (:895-1000)."""
        batch, real, gain, blur = ctx['batch'], ctx['real'], ctx['gain'], ctx['blur_sigma']
        run = self.run_D_semantic if semantic else self.run_D
        tag = '_semantic' if semantic else ''
        loss_gen = 0
        if phase in ('Dmain', 'Dboth'):                                                   # minimise the logits of generated samples
            mode, c_render = self._pick_view(batch, ctx['gen_c'])
            gen, _ = self._render(ctx, mode=mode, update_emas=True)
            if semantic:
                logits = run(self._with_labels(gen, detach_rgb=False), c_render, blur_sigma=blur)
            else:
                logits = run(gen, c_render, blur_sigma=blur, update_emas=True)
            self.report('Loss/scores/fake' + tag, logits); self.report('Loss/signs/fake' + tag, logits.sign())
            loss_gen = F.softplus(logits)
            loss_gen.mean().mul(gain).backward()
        # real input: maximise its logits (main) and / or penalise their input gradient (reg)
        with_r1 = phase in ('Dreg', 'Dboth')
        img = real['image'].detach().requires_grad_(with_r1)
        raw = real['image_raw'].detach().requires_grad_(with_r1)
        if semantic:
            labels = (F.one_hot(batch['mask'].squeeze(1).long(), num_classes=self.G.semantic_channels).permute(0, 3, 1, 2).float()
                      if self.G.data_type == 'seg' else batch['mask'])
            labels_raw = filtered_resizing(labels, size=ctx['nrr'], f=self.resample_filter, filter_mode=self.filter_mode)
            labels, labels_raw = labels.detach().requires_grad_(with_r1), labels_raw.detach().requires_grad_(with_r1)
            inputs = {'image': torch.cat([img, labels], dim=1), 'image_raw': torch.cat([raw, labels_raw], dim=1)}
        else:
            inputs = {'image': img, 'image_raw': raw}
        logits = run(inputs, batch['pose'], blur_sigma=blur)
        self.report('Loss/scores/real' + tag, logits); self.report('Loss/signs/real' + tag, logits.sign())
        loss_real = loss_r1 = 0
        if phase in ('Dmain', 'Dboth'):
            loss_real = F.softplus(-logits)
            self.report('Loss/D/loss' + tag, loss_gen + loss_real)
        if with_r1:
            penalty = self._r1(logits, [inputs['image'], inputs['image_raw']])
            loss_r1 = penalty * (self.r1_gamma / 2)
            self.report('Loss/r1_penalty' + tag, penalty); self.report('Loss/D/reg' + tag, loss_r1)
        (loss_real + loss_r1).mean().mul(gain).backward()

    def calculate_silhouette_loss(self, weight_image, mask):
        """:1005-1022: accumulated ray weight against the label map's foreground."""
        assert weight_image.shape == mask.shape
        return (weight_image - (mask > 0).float()).pow(2).mean() * 10
