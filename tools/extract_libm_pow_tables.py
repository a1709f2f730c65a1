#!/usr/bin/env python3
"""Extract the two data blocks glibc's float64 `pow` reads (sysdeps/ieee754/dbl-64/e_pow_log_data.c `__pow_log_data`, e_exp_data.c
`__exp_data` [UPSTREAM glibc 2.35, the image's libm.so.6; both from ARM's optimized-routines]) from the installed libm and write them
as one C include.

Why: gymnasium writes Acrobot's and Pendulum's squares as `x ** 2`; on a float64 scalar NumPy and CPython both evaluate that through
libm's `pow(x, 2.0)`, which is within 0.52 ulp but NOT always the correctly rounded x * x (0.08 % of arguments differ in the last
bit; verified here).  Reproducing the reference's float64 state bit for bit therefore needs pow's algorithm with pow's own tables.
The blocks are DATA of a third-party dependency, located by their leading constants:
  __pow_log_data = { ln2hi, ln2lo, poly[7], tab[128] = { invc, pad, logc, logctail } }      (ln2hi = 0x1.62e42fefa3800p-1, then ln2lo, then -0.5)
  __exp_data     = { invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5], tab[2 * 128] }   (N = 128, shift = 0x1.8p52)
tests/test_libm_sincos_cpu.py checks the function built on them against libm itself."""
import math
import os
import struct
import sys


def _d(x):
    return struct.pack("<d", x)


def main(out_path, libm="/lib/x86_64-linux-gnu/libm.so.6"):
    b = open(libm, "rb").read()
    # ---- log side
    key = _d(float.fromhex("0x1.62e42fefa3800p-1")) + _d(float.fromhex("0x1.ef35793c76730p-45")) + _d(-0.5)
    p = b.find(key)
    assert p >= 0 and b.find(key, p + 1) < 0, "__pow_log_data not found (or not unique)"
    head = struct.unpack_from("<9d", b, p)                       # ln2hi ln2lo A[0..6]
    tab = struct.unpack_from("<512d", b, p + 72)
    log_rows = []
    for i in range(128):
        invc, pad, logc, tail = tab[4 * i:4 * i + 4]
        assert pad == 0.0, i
        # sanity: logc + logctail is -log(invc) to double-double accuracy's leading part; invc approximates the reciprocal of the subinterval centre
        assert abs(logc + math.log(invc)) < 1e-12, (i, invc, logc)
        log_rows.append((invc, logc, tail))
    assert log_rows[0][0] > 1.0 and log_rows[127][0] < 1.0
    # ---- exp side
    key = _d(float.fromhex("0x1.71547652b82fep0") * 128) + _d(float.fromhex("0x1.8p52"))
    q = b.find(key)
    assert q >= 0 and b.find(key, q + 1) < 0, "__exp_data not found (or not unique)"
    ehead = struct.unpack_from("<8d", b, q)                      # invln2N shift negln2hiN negln2loN C2 C3 C4 C5
    assert abs(ehead[4] - 0.5) < 1e-12 and abs(ehead[5] - 1 / 6) < 1e-12
    etab = struct.unpack_from("<256Q", b, q + 8 * 14)            # (+ exp2_shift, exp2_poly[5])
    for i in range(128):    # sanity: tab[2i+1] + (i << 45) is the bit pattern of 2^(i/128), tab[2i] a tiny tail
        v = struct.unpack("<d", struct.pack("<Q", (etab[2 * i + 1] + (i << 45)) & (2 ** 64 - 1)))[0]
        assert abs(v - 2.0 ** (i / 128)) < 4e-16, (i, v)
        assert abs(struct.unpack("<d", struct.pack("<Q", etab[2 * i]))[0]) < 2e-16

    def w(v):
        return "0x%016xULL" % struct.unpack("<Q", _d(v))[0]

    with open(out_path, "w") as f:
        f.write("/* glibc 2.35 __pow_log_data and __exp_data as float64 bit patterns.  DATA extracted from the image's libm.so.6 by\n"
                "   tools/extract_libm_pow_tables.py.  Layout (words):\n"
                "     [0..8]    ln2hi, ln2lo, A[0..6]                                  (log side: head)\n"
                "     [9..16]   invln2N, shift, negln2hiN, negln2loN, C2, C3, C4, C5  (exp side: head)\n"
                "     [17 + 3i + {0,1,2}]  invc, logc, logctail of subinterval i, i = 0 .. 127\n"
                "     [401 + 2i + {0,1}]   exp tail, exp scale bits of 2^(i/128),  i = 0 .. 127 */\n")
        f.write("  " + ", ".join(w(v) for v in head) + ",\n")
        f.write("  " + ", ".join(w(v) for v in ehead) + ",\n")
        for r in log_rows:
            f.write("  " + ", ".join(w(v) for v in r) + ",\n")
        for i in range(128):
            f.write("  0x%016xULL, 0x%016xULL,\n" % (etab[2 * i], etab[2 * i + 1]))
    print(f"wrote {out_path}: {17 + 384 + 256} words from {libm} @ {p:#x}, {q:#x}")
    main_f32(os.path.join(os.path.dirname(out_path), "nsg_powf_tab.inc"), b, libm)


def main_f32(out_path, b, libm):
    """float32 `powf` (sysdeps/ieee754/flt-32/e_powf.c: __powf_log2_data = { tab[16] = { invc, logc }, poly[5] }, __exp2f_data = { tab[32],
    shift_scaled, poly[3], ... }): Pendulum's `u ** 2` is a float32 scalar power.  Located by structure: 16 (invc, logc) pairs with
    logc = -log2(invc), one of them exactly (1, 0); 32 words that are the bit patterns of 2^(i/32) less i << 47."""
    n = len(b) // 8
    d = struct.unpack_from("<%dd" % n, b, 0)
    u = struct.unpack_from("<%dQ" % n, b, 0)

    def is_pair(k):
        invc, logc = d[k], d[k + 1]
        return 0.5 < invc < 2.0 and abs(logc + math.log2(invc)) < 1e-9

    logs = [k for k in range(n - 40) if d[k] == 1.0 and d[k + 1] == 0.0 and any(
        all(is_pair(k - 2 * j + 2 * m) for m in range(16)) for j in range(16) if k - 2 * j >= 0)]
    starts = sorted({k - 2 * j for k in logs for j in range(16) if k - 2 * j >= 0 and all(is_pair(k - 2 * j + 2 * m) for m in range(16))
                     and not is_pair(k - 2 * j - 2)})
    starts = [k for k in starts if abs(d[k + 36] - 1 / math.log(2)) < 1e-9]      # (log2f's own table is the same shape with FOUR coefficients after it)
    assert len(starts) == 1, starts
    lp = starts[0]
    ltab = d[lp:lp + 32]
    poly = d[lp + 32:lp + 37]
    assert abs(poly[4] - 1 / math.log(2)) < 1e-9, poly           # the linear coefficient of log2(1 + r)
    exps = [k for k in range(n - 40) if u[k] == 0x3ff0000000000000 and all(
        abs(struct.unpack("<d", struct.pack("<Q", (u[k + i] + (i << 47)) & (2 ** 64 - 1)))[0] - 2.0 ** (i / 32)) < 4e-16 for i in range(32))]
    assert len(exps) == 1, exps
    ep = exps[0]
    etab = u[ep:ep + 32]
    shift, c0, c1, c2 = d[ep + 32:ep + 36]
    assert shift == float.fromhex("0x1.8p52") / 32 and abs(c2 - math.log(2)) < 1e-9, (shift, c0, c1, c2)
    with open(out_path, "w") as f:
        f.write("/* glibc 2.35 __exp2f_data and __powf_log2_data as float64 bit patterns.  DATA extracted from the image's libm.so.6 by\n"
                "   tools/extract_libm_pow_tables.py.  Layout (words):\n"
                "     [0..31]   exp2 scale bits of 2^(i/32)      [32] shift / 32      [33..35] C0, C1, C2\n"
                "     [36 + 2i + {0,1}]  invc, logc (= -log2 invc) of subinterval i, i = 0 .. 15      [68..72] A0 .. A4 */\n")
        for i in range(0, 32, 4):
            f.write("  " + ", ".join("0x%016xULL" % v for v in etab[i:i + 4]) + ",\n")
        f.write("  " + ", ".join("0x%016xULL" % struct.unpack("<Q", _d(v))[0] for v in (shift, c0, c1, c2)) + ",\n")
        for i in range(16):
            f.write("  " + ", ".join("0x%016xULL" % struct.unpack("<Q", _d(v))[0] for v in ltab[2 * i:2 * i + 2]) + ",\n")
        f.write("  " + ", ".join("0x%016xULL" % struct.unpack("<Q", _d(v))[0] for v in poly) + ",\n")
    print(f"wrote {out_path}: 73 words from {libm} @ {ep * 8:#x}, {lp * 8:#x}")


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "include", "nsg_pow_tab.inc"))
