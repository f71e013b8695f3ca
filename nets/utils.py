"""`nets/utils.py:11-28` of the reference: (de)normalisation helpers re-exported by the package."""


def denormalize(kps, data_mean, data_std):
    '''kps: (B, T, C)'''
    data_std = data_std.reshape(1, 1, -1)
    data_mean = data_mean.reshape(1, 1, -1)
    return (kps * data_std) + data_mean


def normalize(kps, data_mean, data_std):
    '''kps: (B, T, C)'''
    data_std = data_std.squeeze().reshape(1, 1, -1)
    data_mean = data_mean.squeeze().reshape(1, 1, -1)
    return (kps - data_mean) / data_std
