"""HEALPix is an external library of the reference (not in its tree).  The oracle restates nest2ring / udgrade_ring
from the published algorithm (Gorski et al. 2005); these checks pin that restatement WITHOUT the library:
 * nest2ring is a bijection, the identity at nside = 1, and reproduces the nside = 2 table printed in the HEALPix
   documentation / healpy (`hp.nest2ring(2, np.arange(48))`);
 * geometry: the pixel centre computed in the RING scheme (closed form from the ring index) equals the one computed in
   the NESTED scheme (face, x, y) for every pixel -- two independent formulas of the paper;
 * a parent's NESTED children are its 4 nearest pixel centres of the finer grid; degrade(upgrade(m)) == m; the
   degraded constant map is constant; udgrade_rms / udgrade_mask follow src/dang_util_mod.f90:341-376."""
import numpy as np

import oracle_ffi as O


def _n2r(nside):
    return np.array([O.nest2ring(nside, p) for p in range(12 * nside * nside)])


def _ang_ring(nside, ip):
    """pix2ang_ring: (z, phi) of RING pixel ip (Gorski et al. 2005, eqs. 2-9)."""
    npix, ncap = 12 * nside * nside, 2 * nside * (nside - 1)
    if ip < ncap:
        i = int(0.5 * (1 + np.sqrt(1 + 2 * ip)))
        while 2 * i * (i - 1) > ip:
            i -= 1
        while 2 * (i + 1) * i <= ip:
            i += 1
        j = ip + 1 - 2 * i * (i - 1)
        return 1.0 - i * i / (3.0 * nside * nside), (j - 0.5) * np.pi / (2 * i)
    if ip < npix - ncap:
        q = ip - ncap
        i = q // (4 * nside) + nside
        j = q % (4 * nside) + 1
        s = 0.5 * (1 + ((i + nside) & 1))
        return (2 * nside - i) * 2.0 / (3.0 * nside), (j - s) * np.pi / (2 * nside)
    z, phi = _ang_ring(nside, npix - 1 - ip)
    return -z, 2 * np.pi - phi


def _ang_nest(nside, ip):
    """pix2ang_nest: (z, phi) from the face number and the in-face (x, y) (same paper, section 4.1)."""
    jrll = [2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4]
    jpll = [1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7]
    face, ipf = divmod(ip, nside * nside)
    ix = sum(((ipf >> (2 * b)) & 1) << b for b in range(16))
    iy = sum(((ipf >> (2 * b + 1)) & 1) << b for b in range(16))
    jr = jrll[face] * nside - ix - iy - 1
    if jr < nside:
        nr, z, ks = jr, 1.0 - jr * jr / (3.0 * nside * nside), 0
    elif jr > 3 * nside:
        nr = 4 * nside - jr
        z, ks = -1.0 + nr * nr / (3.0 * nside * nside), 0
    else:
        nr, z, ks = nside, (2 * nside - jr) * 2.0 / (3.0 * nside), (jr - nside) & 1
    jp = (jpll[face] * nr + ix - iy + 1 + ks) // 2
    if jp > 4 * nr:
        jp -= 4 * nr
    if jp < 1:
        jp += 4 * nr
    return z, (jp - (ks + 1) * 0.5) * np.pi / (2 * nr)


def test_nest2ring_bijection_identity_and_known_table():
    assert list(_n2r(1)) == list(range(12))
    # hp.nest2ring(2, np.arange(48))
    assert list(_n2r(2)) == [13, 5, 4, 0, 15, 7, 6, 1, 17, 9, 8, 2, 19, 11, 10, 3, 28, 20, 27, 12, 30, 22, 21, 14, 32, 24,
                             23, 16, 34, 26, 25, 18, 44, 37, 36, 29, 45, 39, 38, 31, 46, 41, 40, 33, 47, 43, 42, 35]
    for nside in (4, 8, 16):
        assert sorted(_n2r(nside)) == list(range(12 * nside * nside))


def test_ring_and_nested_pixel_centres_agree():
    for nside in (1, 2, 4, 8):
        for p in range(12 * nside * nside):
            zr, pr = _ang_ring(nside, O.nest2ring(nside, p))
            zn, pn = _ang_nest(nside, p)
            assert abs(zr - zn) < 1e-14 and abs((pr - pn + np.pi) % (2 * np.pi) - np.pi) < 1e-13, (nside, p)


def test_nested_children_are_the_nearest_fine_pixels():
    def vec(nside, ipring):
        z, phi = _ang_ring(nside, ipring)
        s = np.sqrt(max(0.0, 1 - z * z))
        return np.array([s * np.cos(phi), s * np.sin(phi), z])
    nc, nf = 2, 4
    fine = np.array([vec(nf, q) for q in range(12 * nf * nf)])
    for p in range(12 * nc * nc):
        kids = sorted(O.nest2ring(nf, 4 * p + t) for t in range(4))
        near = sorted(np.argsort(-fine @ vec(nc, O.nest2ring(nc, p)))[:4].tolist())
        assert kids == near, p


def test_udgrade_modes():
    rng = np.random.default_rng(2)
    ni, no = 8, 2
    m = rng.normal(size=12 * ni * ni)
    d = O.udgrade(0, m, ni, no)
    n2r_i, n2r_o = _n2r(ni), _n2r(no)
    r = (ni // no) ** 2
    for q in range(12 * no * no):          # sequential mean over the NESTED children
        tot = 0.0
        for t in range(r):
            tot = tot + m[n2r_i[q * r + t]]
        assert d[n2r_o[q]] == tot / r
    assert np.allclose(O.udgrade(0, O.udgrade(0, d, no, ni), ni, no), d, rtol=1e-15, atol=0)  # degrade(upgrade(x)) == x
    assert np.allclose(O.udgrade(0, np.full(m.size, 3.25), ni, no), 3.25, rtol=0, atol=0)
    rms = rng.uniform(0.5, 2.0, m.size)
    want = np.sqrt(O.udgrade(0, rms * rms, ni, no)) * (no * 1.0 / ni)            # udgrade_rms
    assert np.array_equal(O.udgrade(1, rms, ni, no), want)
    mask = (rng.uniform(size=m.size) > 0.4).astype(float)
    want = np.where(O.udgrade(0, mask, ni, no) < 0.5, 0.0, 1.0)                  # udgrade_mask, threshold 0.5
    assert np.array_equal(O.udgrade(2, mask, ni, no), want)
    bad = m.copy()
    bad[n2r_i[:r]] = -1.6375e30                                                 # a fully bad parent stays bad,
    bad[n2r_i[r]] = -1.6375e30                                                  # a partly bad one averages the good children
    db = O.udgrade(0, bad, ni, no)
    assert db[n2r_o[0]] == -1.6375e30
    tot = 0.0
    for t in range(1, r):
        tot = tot + m[n2r_i[r + t]]
    assert db[n2r_o[1]] == tot / (r - 1)
