"""Shared helpers for the parity tests: golden fixture access and case enumeration."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CROPS = {1: [[1, 2]], 2: [[1, 2], [0, 1]], 3: [[1, 2], [0, 1], [1, 0]]}
_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN, name))
    return _cache[name]


def float_cases(fixture, nds=(1, 2, 3), dts=("f32", "f64")):
    """yield (key, nd, dt, pad, active, crop-or-None, x, w, go_full, out, gx, gw)"""
    d = golden(fixture)
    for nd in nds:
        for dt in dts:
            x, go, w = d["x_%dd_%s" % (nd, dt)], d["go_%dd_%s" % (nd, dt)], d["w_%dd_%s" % (nd, dt)]
            for pad in range(5):
                for active in (0, 1):
                    for ci, crop in enumerate((None, CROPS[nd])):
                        key = "%dd_%s_p%d_a%d_c%d" % (nd, dt, pad, active, ci)
                        yield key, nd, dt, pad, active, crop, x, w, go, d["out_" + key], d["gx_" + key], d["gw_" + key]


def quant_cases(nds=(1, 2, 3)):
    """yield (key, nd, xname, layout, wname, pad, crop, xq, xzp, wq, wzp, out)"""
    d = golden("g2_quant.npz")
    for nd in nds:
        layouts = {1: ["nchw"], 2: ["nchw", "cl"], 3: ["nchw", "cl3d"]}[nd]
        for xname in ("quint8", "qint8", "qint32"):
            xq, xzp = d["xq_%dd_%s" % (nd, xname)], int(d["xzp_%s" % xname])
            for layout in layouts:
                for wname in ("wu8", "wi8"):
                    wq, wzp = d["wq_%dd_%s" % (nd, wname)], int(d["wzp_%s" % wname])
                    for pad in range(5):
                        for ci, crop in enumerate((None, CROPS[nd])):
                            key = "%dd_%s_%s_%s_p%d_c%d" % (nd, xname, layout, wname, pad, ci)
                            yield key, nd, xname, layout, wname, pad, crop, xq, xzp, wq, wzp, d["out_" + key]


def gw16_tol(eps):
    """Bound of a 16-bit weight gradient against the fp64 evaluation, relative to the largest entry (rel_err): the kernels carry the
    sums in fp32 / fp64 and round ONCE to the storage type -- half a unit in the last place, i.e. eps / 2 of the entry (eps =
    torch.finfo(dtype).eps), plus the fp64 -> fp32 -> 16-bit double rounding (2^-17 of that).  A dropped row group or a wrong corner
    shows up far above this; the earlier 2 * eps did not always."""
    return 0.51 * float(eps)


def rel_err(a, ref):
    """max |a - ref| / max(|ref|max, tiny): scale-relative error used for fp tolerances"""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    return np.abs(a - ref).max() / scale
