"""Colour augmentation of the mixed / target image inside the training step (``strong_transform`` ->
``color_jitter`` + ``gaussian_blur``, /root/reference/utils/dacs_transforms.py:11-78)."""


def strong_color(param, data):
    raise NotImplementedError("colour jitter / gaussian blur of strong_transform: pass color_aug_flag=False or a color_aug "
                              "callable to MTMADISE")
