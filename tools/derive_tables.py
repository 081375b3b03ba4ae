#!/usr/bin/env python3
"""Derive the integer parameter tables of the hot path from the reference text.

Run in the build container only (needs /root/reference, which never travels to
the GPU box).  Reads the reference sources AS TEXT (no compilation, no import)
and writes the derived integer data this repo ships:

  * Q15 Gaussian tables  -- DetectKeypoints.cpp:756-801 literals quantised the
    way ``Mask_t = ap_fixed<16,1,AP_RND_ZERO>`` (SIFTParameters.hpp:72) does:
    nearest multiple of 2^-15, ties toward zero.  Only the unique octant
    (r <= c <= centre) is emitted; the tables are symmetric under transpose and
    both flips (checked here).
  * CircleMask row spans -- circle_mask.hpp:16-40 (each row is one run of 1s).
  * XStart / YStart      -- circle_mask.hpp:7-8.
  * MaskAll              -- mask_all.hpp:8-78; checked here to equal the union of
    the nine circles placed at (XStart, YStart), so only that fact is shipped.
  * Cam1 geometry        -- CircleMaskCam1 (29x29) row spans and XStartCam1 / YStartCam1
    (circle_mask.hpp:12-13, :43-71) and MaskAll_81 (mask_all.hpp:161-241).  MaskAll_81 is NOT the
    union of the nine Cam1 circles: it holds 8 more pixels (checked here).  The oracle gets the
    mask itself (81 rows of bits), the product the union fact plus those 8 pixels.

Outputs (identical data, two consumers that must not depend on each other):
  oracle/oracle_tables.h        (CPU oracle, test infrastructure)
  sift_amd/csrc/sift_tables.h   (HIP product)

``--check`` re-derives and compares with the committed files (used by
tests/test_tables.py when the reference is present).
"""
import argparse
import os
import re
import sys
from fractions import Fraction

REF = "/root/reference/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _live_text(path, last_line=None):
    with open(path, "r", errors="replace") as f:
        lines = f.readlines()
    if last_line is not None:
        lines = lines[:last_line]
    return "".join(lines)


def _braced_block(text, decl_regex):
    m = re.search(decl_regex, text)
    if not m:
        raise RuntimeError("declaration not found: " + decl_regex)
    i = text.index("{", m.end() - 1) if text[m.end() - 1] != "{" else m.end() - 1
    depth = 0
    for j in range(i, len(text)):
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i:j + 1]
    raise RuntimeError("unterminated initialiser")


def q15_rnd_zero(lit):
    """double literal -> Q15 integer; nearest, ties toward zero (AP_RND_ZERO)."""
    v = Fraction(float(lit)) * 32768          # exact value of the C double
    fl = v.numerator // v.denominator
    frac = v - fl
    if frac > Fraction(1, 2):
        return fl + 1
    if frac == Fraction(1, 2):
        raise RuntimeError("tie encountered for literal " + lit)
    return fl


def gaussian_tables():
    # live code ends at DetectKeypoints.cpp:1363; the rest is a commented copy
    text = _live_text(os.path.join(REF, "DetectKeypoints.cpp"), 1363)
    out = []
    for n, size in ((1, 5), (2, 11), (3, 19)):
        blk = _braced_block(text, r"Mask_t\s+GaussianKernel%d\s*\[[^\]]*\]\s*\[[^\]]*\]\s*=\s*\{" % n)
        lits = re.findall(r"[0-9]*\.[0-9]+(?:[eE][-+]?[0-9]+)?", blk)
        assert len(lits) == size * size, (n, len(lits))
        k = [[q15_rnd_zero(lits[r * size + c]) for c in range(size)] for r in range(size)]
        for r in range(size):
            for c in range(size):
                assert k[r][c] == k[c][r] == k[size - 1 - r][c] == k[r][size - 1 - c], (n, r, c)
        out.append(k)
    return out


def int_list(blk):
    return [int(t) for t in re.findall(r"\d+", blk)]


def circle_tables():
    text = _live_text(os.path.join(REF, "circle_mask.hpp"))
    xs = int_list(_braced_block(text, r"ap_uint<8>\s+XStart\s*\[9\]\s*=\s*\{"))
    ys = int_list(_braced_block(text, r"ap_uint<8>\s+YStart\s*\[9\]\s*=\s*\{"))
    cm = int_list(_braced_block(text, r"ap_uint<8>\s+CircleMask\s*\[25\*25\]\s*=\s*\{"))
    assert len(xs) == 9 and len(ys) == 9 and len(cm) == 625
    assert set(cm) <= {0, 1}
    spans = []
    for r in range(25):
        row = cm[r * 25:(r + 1) * 25]
        ones = [i for i, v in enumerate(row) if v]
        assert ones == list(range(ones[0], ones[-1] + 1)), "row is not a single run"
        spans.append((ones[0], ones[-1] + 1))
    return xs, ys, cm, spans


def mask_all_table():
    text = _live_text(os.path.join(REF, "mask_all.hpp"))
    blk = _braced_block(text, r"ap_uint<16>\s+MaskAll\s*\[71\]\s*\[71\]\s*=\s*\{")
    v = int_list(blk)
    assert len(v) == 71 * 71 and set(v) <= {0, 1}
    return [v[r * 71:(r + 1) * 71] for r in range(71)]


def cam1_tables():
    text = _live_text(os.path.join(REF, "circle_mask.hpp"))
    xs = int_list(_braced_block(text, r"ap_uint<8>\s+XStartCam1\s*\[9\]\s*=\s*\{"))
    ys = int_list(_braced_block(text, r"ap_uint<8>\s+YStartCam1\s*\[9\]\s*=\s*\{"))
    cm = int_list(_braced_block(text, r"ap_uint<8>\s+CircleMaskCam1\s*\[29\*29\]\s*=\s*\{"))
    assert len(xs) == 9 and len(ys) == 9 and len(cm) == 841 and set(cm) <= {0, 1}
    spans = []
    for r in range(29):
        row = cm[r * 29:(r + 1) * 29]
        ones = [i for i, v in enumerate(row) if v]
        assert ones == list(range(ones[0], ones[-1] + 1)), "row is not a single run"
        spans.append((ones[0], ones[-1] + 1))
    mt = _live_text(os.path.join(REF, "mask_all.hpp"))
    v = int_list(_braced_block(mt, r"ap_uint<16>\s+MaskAll_81\s*\[81\]\s*\[81\]\s*=\s*\{"))
    assert len(v) == 81 * 81 and set(v) <= {0, 1}
    ma = [v[r * 81:(r + 1) * 81] for r in range(81)]
    cnt = [[0] * 81 for _ in range(81)]
    for n in range(9):
        for i in range(29):
            for j in range(29):
                if cm[i * 29 + j]:
                    cnt[ys[n] + i][xs[n] + j] += 1
    assert max(map(max, cnt)) == 2, "a Cam1 pixel lies in more than two circles"
    assert all(ma[r][c] for r in range(81) for c in range(81) if cnt[r][c]), "MaskAll_81 misses a union pixel"
    extra = [(r, c) for r in range(81) for c in range(81) if ma[r][c] and not cnt[r][c]]
    assert len(extra) == 8
    return xs, ys, spans, ma, extra


def derive():
    g = gaussian_tables()
    xs, ys, cm, spans = circle_tables()
    ma = mask_all_table()
    union = [[0] * 71 for _ in range(71)]
    for n in range(9):
        for i in range(25):
            for j in range(25):
                if cm[i * 25 + j]:
                    union[ys[n] + i][xs[n] + j] = 1
    assert union == ma, "MaskAll is not the union of the nine circles"
    assert all(row[70] == 0 for row in ma)
    return g, xs, ys, spans


def emit(prefix, guard, role, qual="static const"):
    g, xs, ys, spans = derive()
    L = []
    L.append("/* GENERATED by tools/derive_tables.py -- do not edit.")
    L.append(" * %s" % role)
    L.append(" * Q15 Gaussian tables: unique octant K[r][c], r <= c <= centre, row by row;")
    L.append(" * the full table is K[r][c] = OCT[min(r',c')][max(r',c')] with r' = min(r, n-1-r).")
    L.append(" * Derived from the literals at DetectKeypoints.cpp:756-801 quantised as")
    L.append(" * ap_fixed<16,1,AP_RND_ZERO> (SIFTParameters.hpp:72). */")
    L.append("#ifndef %s" % guard)
    L.append("#define %s" % guard)
    for idx, k in enumerate(g, 1):
        n = len(k)
        o = n // 2
        vals = [k[r][c] for r in range(o + 1) for c in range(r, o + 1)]
        L.append("/* %dx%d: sum=%d max=%d */" % (n, n, sum(map(sum, k)), k[o][o]))
        L.append("#define %s_G%d_SIZE %d" % (prefix, idx, n))
        L.append(qual + " unsigned short %s_G%d_OCT[%d] = {%s};"
                 % (prefix, idx, len(vals), ",".join(map(str, vals))))
    L.append("/* circle_mask.hpp:7-8 */")
    L.append(qual + " unsigned char %s_XSTART[9] = {%s};" % (prefix, ",".join(map(str, xs))))
    L.append(qual + " unsigned char %s_YSTART[9] = {%s};" % (prefix, ",".join(map(str, ys))))
    L.append("/* circle_mask.hpp:16-40: row i of the 25x25 mask is 1 on [lo,hi) */")
    L.append(qual + " unsigned char %s_CIRCLE_LO[25] = {%s};" % (prefix, ",".join(str(a) for a, _ in spans)))
    L.append(qual + " unsigned char %s_CIRCLE_HI[25] = {%s};" % (prefix, ",".join(str(b) for _, b in spans)))
    L.append("/* mask_all.hpp:8-78 equals the union of the nine circles at (XSTART,YSTART)")
    L.append(" * (checked by the generator); it is rebuilt from that at start-up. */")
    xs1, ys1, spans1, ma1, extra1 = cam1_tables()
    L.append("/* Cam1 geometry: circle_mask.hpp:12-13 */")
    L.append(qual + " unsigned char %s_XSTART_CAM1[9] = {%s};" % (prefix, ",".join(map(str, xs1))))
    L.append(qual + " unsigned char %s_YSTART_CAM1[9] = {%s};" % (prefix, ",".join(map(str, ys1))))
    L.append("/* circle_mask.hpp:43-71: row i of the 29x29 mask is 1 on [lo,hi) */")
    L.append(qual + " unsigned char %s_CIRCLE_CAM1_LO[29] = {%s};" % (prefix, ",".join(str(a) for a, _ in spans1)))
    L.append(qual + " unsigned char %s_CIRCLE_CAM1_HI[29] = {%s};" % (prefix, ",".join(str(b) for _, b in spans1)))
    if prefix == "ORC":
        L.append("/* mask_all.hpp:161-241: MaskAll_81, row r = bits 0..80 of three words (bit c of word c / 32) */")
        rows = []
        for r in range(81):
            w = [0, 0, 0]
            for c in range(81):
                if ma1[r][c]:
                    w[c // 32] |= 1 << (c % 32)
            rows.append("{0x%08xu,0x%08xu,0x%08xu}" % tuple(w))
        L.append(qual + " unsigned int %s_MASKALL81[81][3] = {%s};" % (prefix, ",".join(rows)))
    else:
        L.append("/* mask_all.hpp:161-241: MaskAll_81 = the union of the nine Cam1 circles + these 8 pixels {row, column}")
        L.append(" * (checked by the generator) */")
        L.append(qual + " unsigned char %s_MASKALL81_EXTRA[8][2] = {%s};" % (prefix, ",".join("{%d,%d}" % e for e in extra1)))
    L.append("#endif")
    return "\n".join(L) + "\n"


TARGETS = (
    ("oracle/oracle_tables.h", "ORC", "ORACLE_TABLES_H",
     "CPU oracle copy (test infrastructure; never linked into the product).", "static const"),
    ("sift_amd/csrc/sift_tables.h", "SIFT", "SIFT_TABLES_H",
     "HIP product copy (C++ only: constexpr so kernels fold the values into immediates).", "static constexpr"),
)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        return 0
    rc = 0
    for rel, prefix, guard, role, qual in TARGETS:
        txt = emit(prefix, guard, role, qual)
        path = os.path.join(ROOT, rel)
        if a.check:
            with open(path) as f:
                if f.read() != txt:
                    print("MISMATCH: " + rel)
                    rc = 1
        else:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(txt)
            print("wrote " + rel)
    return rc


if __name__ == "__main__":
    sys.exit(main())
