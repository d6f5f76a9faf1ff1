"""Host-side twin of the keyed random streams (csrc/dx_rng.h): Philox4x32-10, uniform2 / uniform3 and the
reference's rand_normal (src/dang_util_mod.f90:100-110).  Used where the chain logic runs on the host
(full-sky index mode, step-size tuner, band-gain fit); the per-pixel chains draw on the device."""
import math

M32 = 0xFFFFFFFF
GLOBAL_PIX = 0xFFFFFFFFFF  # pixel label of sky-wide draws (no real pixel uses it)


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def _words(seed, stream, pix, draw):
    ctr = (pix & M32, (draw ^ (((pix >> 32) & M32) << 16)) & M32, stream & M32, (stream >> 32) & M32)
    return philox4x32_10(ctr, (seed & M32, (seed >> 32) & M32))


def _u53(hi, lo):
    return (float(((hi << 32) | lo) >> 11) + 0.5) * (1.0 / 9007199254740992.0)


def uniform2(seed, stream, pix, draw):
    o = _words(seed, stream, pix, draw)
    return _u53(o[0], o[1]), _u53(o[2], o[3])


def uniform3(seed, stream, pix, draw):
    o = _words(seed, stream, pix, draw)
    return _u53(o[0], o[1]), (o[2] + 0.5) / 4294967296.0, (o[3] + 0.5) / 4294967296.0


def rand_normal(mean, stdev, u1, u2):
    r = math.sqrt(-2.0 * math.log(u1))
    theta = 2.0 * math.pi * u2
    return mean + stdev * r * math.sin(theta)


def eval_normal_prior(prop, mean, std):
    """src/dang_util_mod.f90:112-121"""
    var = std * std
    return math.exp(-((prop - mean) * (prop - mean)) / (2 * var)) / (std * math.sqrt(2.0 * math.pi))
