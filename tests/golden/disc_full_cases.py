"""Config 3's discriminators at their real size: constructor arguments, synthetic inputs and the tile statistics the golden keeps — shared by
tests/golden/make_golden.py (group ``discriminator_full``, reference modules) and tests/test_discriminator.py (this package's).  torch only."""
import torch

CASES = (('d', 3), ('dsem', 9))          # (tag, image channels): D on the image, D_semantic on image + 6 label channels


def full_discriminator_kwargs(img_channels):
    """Config 3's discriminators at their real size (train.py:289-318, 509-512; training_loop.py:308 for the label-aware one)."""
    return dict(class_name='training.dual_discriminator.DualDiscriminator', c_dim=25, img_resolution=512, img_channels=img_channels, channel_base=32768, channel_max=512,
                num_fp16_res=4, conv_clamp=256, disc_c_noise=0, block_kwargs=dict(freeze_layers=0), mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=2))


def full_discriminator_inputs(img_channels, n=2):
    g = torch.Generator().manual_seed(41 + img_channels)
    img = torch.rand(n, img_channels, 512, 512, generator=g) * 2 - 1
    raw = torch.rand(n, img_channels, 128, 128, generator=g) * 2 - 1
    if img_channels > 3:                                       # label channels of a real sample are one-hot (loss.py:946-949): make them so at full size
        lab = torch.randint(0, img_channels - 3, [n, 1, 512, 512], generator=g)
        img[:, 3:] = torch.nn.functional.one_hot(lab.squeeze(1), img_channels - 3).permute(0, 3, 1, 2).float()
    c = torch.randn(n, 25, generator=g)
    return img, raw, c


def tile_stats(t, tile=32):
    """Per-tile sum and abs-max of a [N,C,H,W] field: every element moves a recorded number."""
    n, ch, h, w = t.shape
    v = t.detach().double().reshape(n, ch, h // tile, tile, w // tile, tile)
    return v.sum(dim=(3, 5)), v.abs().amax(dim=(3, 5))
