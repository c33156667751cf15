#!/usr/bin/env python3
"""Could another limb layout make the p384 ladder 27 % faster?  (var_p384 >= 3.0e7 /s "or the model's
multiply-add + carry counts for the candidate layouts".)  Pure arithmetic on the structure of ecgpu_field.h — no GPU.

For an unsaturated Montgomery field of NL limbs of B bits (R = 2^(NL B), p = -1 mod 2^B so that p' = 1, the sparse signed
reduction rows of p384: u p = -u + u 2^32 - u 2^96 - u 2^128 + u 2^384) one multiplication costs, in issue slots of 4 cycles
(profiles/r01/isa_issue_rates.txt: v_mad_u64_u32 and every other VOP3 / 64-bit instruction 1 slot, 32-bit VOP2 1/2):

    products        NL^2 multiply-adds (squaring: NL (NL + 1) / 2, the doubled cross terms need NL shifted operands: + NL / 2)
    reduction rows  NL x (4 multiply-adds + mask 1/2 + 64-bit arithmetic shift 1 + 64-bit add 1)
    carry pass      NL x (mask 1/2 + shift 1 + add 1)
    operand traffic moves into / out of the 64-bit column registers: ~ NL / 2 (measured: ~10 % v_mov in these kernels)

and the layout must leave HEADROOM: a 64-bit signed column receives up to NL products of two limbs of magnitude ma, mb
(limb bound ~ m 2^B), so NL ma mb 2^(2B) < 2^63: product limit ma mb <= 2^(63 - 2B) / NL.  The group law as written
(ecgpu_point.h, a = -3) needs 23 (o.y = yy_p (4) * yy_m (5) + xx3_m_zz3 (1) * bxz3 (3)); below that, operands have to be
normalised first (one carry pass each, 2.5 NL slots) — the model counts how many of the formulas' products exceed the limit.

A second headroom is the VALUE magnitude: Montgomery reduction needs a b < R p, i.e. for lazily reduced operands of value
magnitudes va, vb (value < v 2p) va vb < R / 4p = 2^(NL B - 386).  The ladder keeps its coordinates at value magnitude <= 11
(ecgpu_point.h JV: sums and differences are not reduced below 2p between multiplications), so it needs va vb up to 121:
2^19 of room at 15 x 27, 2^6 = 64 at 14 x 28, 2^4 = 16 at 13 x 30.  Below 121 every coordinate would have to be brought
back under 2p before it is multiplied — a conditional subtraction chain per operand (~ 3 NL slots), charged below as
`value fixes` for the products whose value magnitudes exceed the room (11 x 11, 11 x 2, ... from the source).

The ladder is 4 doublings (3M + 5S, dbl-2001-b) + 1 mixed addition (8M + 3S, madd-2004-hmv) per digit; additions and
subtractions are NL 32-bit adds (1/2 slot each) and are charged per formula from the source (doubling 14, addition 9).
"""
import sys

P = 2 ** 384 - 2 ** 128 - 2 ** 96 + 2 ** 32 - 1

# products of the two formulas with the limb magnitudes of their operands as the code has them today
# (ecgpu_point.h jac_dbl a = -3 and jac_madd): (ma, mb) per multiplication / squaring
DBL = [(1, 1), (1, 1), (1, 1), (3, 2), (3, 3), (3, 6), (2, 2), (1, 1)]                   # delta, gamma, beta, alpha, alpha3^2, alpha3*(..), (Y+Z)^2, gamma^2
MADD = [(1, 1), (1, 1), (1, 1), (1, 1), (2, 1), (1, 1), (1, 1), (1, 1), (2, 2), (2, 3), (2, 1)]  # zz1, U2, t, Z3, S2, HH, V, HHH, r^2, r*(V-X3), Y1*HHH


def cost(nl, bits, split=False):
    assert (P + 1) % (1 << bits) == 0 or (P % (1 << bits)) == (1 << bits) - 1, "p = -1 mod 2^B needed for p' = 1"
    limit = (1 << (63 - 2 * bits)) // nl                       # signed columns
    if split:                                                  # two accumulators per column: each takes half the products
        limit = (1 << (63 - 2 * bits)) // ((nl + 1) // 2)
    mul = nl * nl + nl * (4 + 0.5 + 2) + nl * 2.5 + nl / 2
    sqr = nl * (nl + 1) / 2 + nl / 2 + nl * (4 + 0.5 + 2) + nl * 2.5 + nl / 2
    if split:
        mul += nl * 2 - 1                                      # joining the two halves of every column: 64-bit adds
        sqr += nl * 2 - 1
    norm = nl * 2.5
    add = nl * 0.5

    def formula(prods, nm, ns, nadd):
        over = sum(1 for ma, mb in prods if ma * mb > limit)   # one operand normalised first
        return nm * mul + ns * sqr + nadd * add + over * norm, over

    d, dov = formula(DBL, 3, 5, 14)
    a, aov = formula(MADD, 8, 3, 9)
    vroom = nl * bits - 386                                    # log2 of R / 4p
    vfix = 0
    if (1 << vroom) < 121:                                     # every stored coordinate re-enters a product at value magnitude ~11
        vfix = 3 * nl * (3 * 4 + 3)                            # three coordinates per doubling and per addition result
    digit = 4 * d + a + vfix
    return dict(nl=nl, bits=bits, split=split, limit=limit, mul=mul, sqr=sqr, dbl=d, madd=a, digit=digit, extra_norms=(dov, aov),
                vroom=vroom, vfix=vfix)


def karatsuba_and_windows():
    """Round 4's review, item 8: "model the multiplication, not only the limb width" — one level of Karatsuba on the 15 limbs (8 + 7)
    and signed 5-bit windows for the variable-time ladder, in the same slot units as cost() above; build whichever comes out >= 5 %
    ahead.  Neither does."""
    r = cost(15, 27)
    nl, mul, sqr = 15, r["mul"], r["sqr"]
    print("\n--- one level of Karatsuba, 15 = 8 + 7 limbs (a = a0 + a1 2^216) ---")
    # subtractive form on the signed columns: a b = z0 + z2 2^432 + (z0 + z2 - (a0 - a1)(b0 - b1)) 2^216
    prods = 8 * 8 + 7 * 7 + 8 * 8
    limb_subs = 2 * 8 * 0.5                  # a0 - a1, b0 - b1: 32-bit subtractions
    # z0 (15 columns) and z2 (13) are needed twice, at their own position and shifted by 8 limbs: the middle product is accumulated
    # onto the overlapping result columns by its own multiply-adds (free), the shifted copies are 64-bit additions
    col_adds = 15 + 13
    extra = limb_subs + col_adds
    saved = nl * nl - prods
    kmul = mul - saved + extra
    print("products %d against %d (-%d multiply-adds); + %d 64-bit column additions + %d limb subtractions (%.0f slots)" % (prods, nl * nl, saved, col_adds, 16, extra))
    print("multiplication %.1f -> %.1f slots (%.1f %%)" % (mul, kmul, 100 * (kmul / mul - 1)))
    sq_prods = 36 + 28 + 36                  # three half-size squarings
    ksqr = sqr - (nl * (nl + 1) / 2 - sq_prods) + limb_subs / 2 + col_adds
    print("squaring       %.1f -> %.1f slots (%.1f %%): %d products against %d do not pay for the same %d column additions -> squarings stay schoolbook" % (
        sqr, ksqr, 100 * (ksqr / sqr - 1), sq_prods, nl * (nl + 1) // 2, col_adds))
    nmul = 4 * 3 + 8                         # multiplications per digit (4 doublings of 3M + 5S, one mixed addition of 8M + 3S)
    gain = nmul * (mul - kmul)
    print("per digit: %d multiplications x %.1f slots = %.0f of %.0f slots = %.1f %%" % (nmul, mul - kmul, gain, r["digit"], 100 * gain / r["digit"]))
    # headroom: the middle product multiplies DIFFERENCES (limb magnitude ma + mb each side), 8 products per column
    lim_k = (1 << (63 - 2 * 27)) // (8 * 4 + 15)
    dbl_muls = [DBL[2], DBL[3], DBL[5]]                                   # beta, alpha, alpha3 * (4 beta - X3): the multiplications of the doubling
    madd_muls = [m for i, m in enumerate(MADD) if i not in (0, 5, 8)]     # all but zz1, HH, r^2
    over = 4 * sum(1 for a, b in dbl_muls if a * b > lim_k) + sum(1 for a, b in madd_muls if a * b > lim_k)
    print("headroom: a signed column takes 15 products of magnitudes ma mb <= %d today; the middle product's operands are differences of two\n"
          "  halves (magnitude x 2 each -> x 4 per product) on columns that also carry z0 + z2: ma mb <= %d -> %d multiplications per digit\n"
          "  (alpha3 (3) x (4 beta - X3) (6) of every doubling) need an operand normalised first (%.1f slots each): %.0f of the %.0f slots gained"
          % (r["limit"], lim_k, over, 2.5 * nl, over * 2.5 * nl, gain))
    net = gain - over * 2.5 * nl
    print("=> Karatsuba: %.1f %% of the ladder before, %.1f %% after the normalisations (and z0, z2 held apart: 28 more live 64-bit columns in a\n"
          "   kernel that already spills 146 registers at two waves per SIMD): not built" % (100 * gain / r["digit"], 100 * net / r["digit"]))

    print("\n--- signed 5-bit windows for the variable-time ladder ---")
    jadd = 11 * mul + 5 * sqr                # Jacobian + Jacobian (add-2007-bl) for the table
    inv = 6.0e4 - 7 * jadd - 7 * 5 * mul     # what is left of the measured ~6e4 slots of table construction: the inversion by division steps
    def ladder(wbits, ndig, entries):
        table = (entries - 1) * jadd + inv + (entries - 1) * 5 * mul
        return ndig * (wbits * r["dbl"] + r["madd"]) + table, table
    t4, tab4 = ladder(4, 97, 8)
    t5, tab5 = ladder(5, 77, 16)
    print("4-bit: 97 digits x (4 doublings + 1 mixed addition) = %.4g slots + table of  8 affine entries %.3g = %.4g" % (t4 - tab4, tab4, t4))
    print("5-bit: 77 digits x (5 doublings + 1 mixed addition) = %.4g slots + table of 16 affine entries %.3g = %.4g  (%.1f %%)" % (t5 - tab5, tab5, t5, 100 * (t5 / t4 - 1)))
    print("  20 mixed additions fewer (%.3g slots), 1 doubling more (385 against 388 - 3: none), 8 table entries more (%.3g slots);" % (20 * r["madd"], tab5 - tab4))
    print("  the table lives in HBM scratch per lane (VarTabHbm, ecgpu_var.h: 8 x 2 x 64 B = 1 KB per lane, 131,072 resident lanes = 134 MB);\n"
          "  16 entries double it and the build's stores, the gather per digit stays one 128-byte entry")
    print("=> 5-bit windows: %.1f %% fewer slots, before the scratch gathers: var_p384 43.4-44.5 ms -> ~%.1f at best, target 41: not built" % (100 * (1 - t5 / t4), 43.9 * t5 / t4))
    print("   (a width-5 NAF would need ~64 additions, but its positions depend on the scalar: in a wave of 64 scalars some lane adds at\n"
          "    nearly every position, so every lane walks every addition — the fixed windows are the SIMT-friendly form)")


def main():
    if "--karatsuba-windows" in sys.argv:
        return karatsuba_and_windows()
    rows = [cost(15, 27), cost(14, 28), cost(13, 30, split=True), cost(13, 30)]
    base = rows[0]["digit"]
    print("%-22s %6s %7s %8s %8s %9s %9s %8s %10s %8s  %s" % ("layout", "limit", "R/4p", "mul", "sqr", "doubling", "addition", "v.fixes", "per digit", "vs 15x27",
                                                                 "extra normalisations (dbl, add)"))
    for r in rows:
        name = "%d x %d%s" % (r["nl"], r["bits"], " split columns" if r["split"] else "")
        ok = "" if r["limit"] >= 1 else "  (no headroom at all: unusable)"
        print("%-22s %6d %7s %8.1f %8.1f %9.0f %9.0f %8.0f %10.0f %7.1f%%  %s%s" % (name, r["limit"], "2^%d" % r["vroom"], r["mul"], r["sqr"], r["dbl"], r["madd"],
                                                                               r["vfix"], r["digit"], 100 * (r["digit"] / base - 1), r["extra_norms"], ok))
    need = 2.36e7 / 3.0e7 - 1
    print("\nneeded for 3.0e7 /s from the measured 2.36e7 /s at the same issue efficiency: %.1f %% fewer slots per digit" % (100 * need))
    print("measured check of the model: k_var_base<P384Params> executes 1.566e6 VALU instructions per scalar x 0.916 slots = 1.43e6 slots;")
    print("the model's 97 digits x %.0f = %.3g slots + table construction (7 additions + inversion + 7 x 5 multiplications ~ 6e4)" % (base, 97 * base))


if __name__ == "__main__":
    sys.exit(main())
