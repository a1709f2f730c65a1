#!/usr/bin/env python3
"""Extract glibc's `__sincostab` (sysdeps/ieee754/dbl-64/sincostab.c [UPSTREAM glibc 2.35, the image's libm.so.6]: sin(k/128) and
cos(k/128), k = 0 .. 109, each as a high and a low double) from the installed libm and write it as a C include.

Why: the base MDPs' float64 trajectories go through `np.sin` / `np.cos`, which NumPy 2.2 resolves to libm's `sin` / `cos`
(verified on this image: 0 differing results in 300 000 samples).  Reproducing the reference's states BIT FOR BIT therefore needs
libm's algorithm with libm's own table; recomputing the table from exact arithmetic reproduces 93 of its 110 entries - the low
words of the other 17 are not the correctly rounded remainders.  The table is DATA of a third-party dependency, located by its
known entries 4..7 (sin(1/128), its low word, cos(1/128), its low word).  tests/test_libm_sincos_cpu.py checks the functions built
on it against libm itself."""
import os
import struct
import sys


def main(out_path, libm="/lib/x86_64-linux-gnu/libm.so.6"):
    b = open(libm, "rb").read()
    import math
    key = struct.pack("<d", math.sin(1 / 128))                  # entry 1's high word of the sine (libm's own correctly rounded value)
    hits = []
    p = b.find(key)
    while p >= 0:
        if p >= 32 and b[p - 32:p - 32 + 24] == struct.pack("<ddd", 0.0, 0.0, 1.0):      # entry 0: sin 0 = 0 + 0, cos 0 = 1 (+ 0)
            hits.append(p - 32)
        p = b.find(key, p + 1)
    assert len(hits) >= 1, "__sincostab not found"
    tab = struct.unpack_from("<440d", b, hits[0])
    for k in range(110):     # sanity: high words are libm's own sin / cos of k/128
        assert tab[4 * k] == math.sin(k / 128) and tab[4 * k + 2] == math.cos(k / 128), k
    with open(out_path, "w") as f:
        f.write("/* glibc 2.35 __sincostab: sin(k/128) = x[4k] + x[4k+1], cos(k/128) = x[4k+2] + x[4k+3], k = 0 .. 109.  DATA extracted from the\n"
                "   image's libm.so.6 by tools/extract_libm_sincos_table.py (440 float64 values as bit patterns). */\n")
        for k in range(110):
            f.write("  " + ", ".join("0x%016xULL" % struct.unpack("<Q", struct.pack("<d", v))[0] for v in tab[4 * k:4 * k + 4]) + ",\n")
    print(f"wrote {out_path}: 440 words from {libm} @ {hits[0]:#x}")


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "include", "nsg_sincos_tab.inc"))
