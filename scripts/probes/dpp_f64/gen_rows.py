"""Generates the whitening's fma chains as inline asm with the W entry of every fma taken from a lane of a VGPR pair by DPP
row_newbcast (v_fmac_f64_dpp: gfx90a+ "DP ALU DPP").  W packed lower-triangular, entry e = i (i + 1) / 2 + j, lives in lane e % 16 of
every row of register pair Wr[e / 16].  G rows per asm block, their chains interleaved (independent accumulators).
usage: python gen_rows.py <D> <G> > rows_<D>_<G>.inc"""
import sys
D = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = []
for i0 in range(0, D, G):
    rows = list(range(i0, min(i0 + G, D)))
    maxj = max(rows)
    wregs = sorted(set((i * (i + 1) // 2 + j) // 16 for i in rows for j in range(i + 1)))
    nacc = len(rows)
    # operands: %0..%(nacc-1) accs; then rr[0..maxj]; then W regs
    rr_idx = {j: nacc + j for j in range(maxj + 1)}
    w_idx = {r: nacc + maxj + 1 + n for n, r in enumerate(wregs)}
    s = '"s_nop 1\\n\\t"\n'
    for j in range(maxj + 1):
        for a, i in enumerate(rows):
            if j <= i:
                e = i * (i + 1) // 2 + j
                s += '            "v_fmac_f64_dpp %%%d, %%%d, %%%d row_newbcast:%d row_mask:0xf bank_mask:0xf\\n\\t"\n' % (a, w_idx[e // 16], rr_idx[j], e % 16)
    accs = ', '.join('"+v"(acc%d)' % a for a in range(nacc))
    ops = ', '.join(['"v"(rr[%d])' % j for j in range(maxj + 1)] + ['"v"(Wr[%d])' % r for r in wregs])
    decl = ' '.join('double acc%d = -0.0;' % a for a in range(nacc))
    qs = ' '.join('q = (%d == 0) ? acc%d * acc%d : fma(acc%d, acc%d, q);' % (i, a, a, a, a) for a, i in enumerate(rows))
    out.append('        { %s\n          asm(%s            : %s : %s);\n          %s }' % (decl, s, accs, ops, qs))
print('\n'.join(out))
