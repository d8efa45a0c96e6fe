#!/usr/bin/env python3
"""Generates demc.jl_amd/csrc/demcz_kernels_ps2d.h (two chains to a wave) from demcz_kernels_ps2.h by substitution: every
statement of the one-chain kernel stays, the places where "the wave's chain" becomes "this lane's chain of the wave's two" are
rewritten (each substitution asserts that its source text is there exactly once, so a change to the one-chain kernel that this
script does not know how to carry over stops it).  `--check`: exit status 0 iff the committed file is what would be generated
(tests/test_abi.py).   usage: python scripts/gen_ps2d.py [--check]"""
import re
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "demc.jl_amd" / "csrc"
_s = (CSRC / "demcz_kernels_ps2.h").read_text()
_a = _s.index('template <int TARGET, int D, bool LIVE, bool TEMPER>\n__global__ void __launch_bounds__(64 * (PS_CHAINS + (LIVE ? 1 : 0)), 3) window_kernel_ps2')
body = _s[_a:]
body = body[:body.rindex('}  // namespace demcz')]
def sub(old,new,count=1):
    global body
    assert body.count(old)>=1, old[:80]
    if count==0: body=body.replace(old,new)
    else:
        assert body.count(old)==count, (body.count(old), old[:80])
        body=body.replace(old,new)

sub('window_kernel_ps2(const WindowParams P)','window_kernel_ps2d(const WindowParams P)')
sub('    constexpr int R = PS2_R;\n','    constexpr int R = PS2_R;\n    constexpr int NCH = 2;                                 // chains of a wave: lanes 0..31 run one, lanes 32..63 the other\n')
sub('raw[PS_CHAINS][PS2_SLOTS][1024];','raw[PS_CHAINS][PS2_SLOTS][NCH * 1024];')
sub('double sdelta[PS_CHAINS][SDN];','double sdelta[PS_CHAINS][NCH][SDN];')
sub('ctab[PS_CHAINS][64 * CR];       // row l: lane l\'s candidate (rows 32..63 shadow 0..31)','ctab[PS_CHAINS][64 * CR];       // row l: lane l\'s candidate (rows 0..31: the first chain\'s tree, 32..63: the second\'s)')
sub('pub_rows[LIVE ? PS_CHAINS * PS_PUB * D : 1];','pub_rows[LIVE ? PS_CHAINS * PS_PUB * NCH * D : 1];')
# publisher
sub('''            const bool pl = lane < PS_CHAINS * D;
            const int cw = pl ? lane / D : 0, pp = pl ? lane % D : 0;
            const int64_t cl = (int64_t)bxs * PS_CHAINS + cw;''','''            const bool pl = lane < PS_CHAINS * NCH * D;
            const int cw = pl ? lane / (NCH * D) : 0, ph = pl ? (lane / D) % NCH : 0, pp = pl ? lane % D : 0;
            const int64_t cl = ((int64_t)bxs * PS_CHAINS + cw) * NCH + ph;''')
sub('if (ready) v = pub_rows[(cw * PS_PUB + (int)(done % PS_PUB)) * D + pp];','if (ready) v = pub_rows[((cw * PS_PUB + (int)(done % PS_PUB)) * NCH + ph) * D + pp];')
sub('if (ready && pp == 0) __hip_atomic_store(&pub_done[cw], done,','if (ready && pp == 0 && ph == 0) __hip_atomic_store(&pub_done[cw], done,')
sub('constexpr int NCHR = 1;                                // chains of a chain wave','constexpr int NCHR = NCH;                              // chains of a chain wave')
sub('constexpr bool HRING = LIVE && (PS2_HRING_ONE != 0);','constexpr bool HRING = LIVE && (PS2_HRING_TWO != 0);')
sub('const int hkk = w, hr = lane;         // HRING: its chain among the workgroup\'s, its row of the pass','const int hkk = w * NCH + hh, hr = l5; // HRING: its chain among the workgroup\'s, its row of the pass')
# chain index
sub('''    const int64_t c = (int64_t)bxs * PS_CHAINS + w;
    if (c >= P.N) {
        wave_store_counts(P, c, 0u, 0u);
        leave();
        return;
    }''','''    const int64_t wv = (int64_t)bxs * PS_CHAINS + w;       // this wave among the chain waves; its chains: NCH * wv, NCH * wv + 1
    if (wv * NCH >= P.N) {
        wave_store_counts(P, wv, 0u, 0u);
        leave();
        return;
    }
    const int hh = lane >> 5, l5 = lane & 31;               // which of the wave's chains this lane works for; its place in that half
    const bool act = wv * NCH + hh < P.N;                  // (an odd population: the last wave's second half shadows the last chain
    const int64_t c = act ? wv * NCH + hh : P.N - 1;       //  -- it reads what that chain reads and writes nothing)
    const unsigned long long actm = (wv * NCH + 1 < P.N) ? ~0ull : 0xffffffffull;''')
sub('''            if (P.safe_X) {
                if (lane < D) P.safe_X[c + P.N * lane] = P.Xcur[c + P.N * lane];
                if (lane == 0) P.safe_lp[c] = P.lpcur[c];
            }''','''            if (P.safe_X && act) {
                if (l5 < D) P.safe_X[c + P.N * l5] = P.Xcur[c + P.N * l5];
                if (l5 == 0) P.safe_lp[c] = P.lpcur[c];
            }''')
sub('double* const sd_w = &sdelta[w][0];','double* const sd_w = &sdelta[w][hh][0];')
sub('''    // node of the tree of outcomes: nn = 0 is the state itself (lanes 0 and 32), 1..31 the nodes (lanes 32..63 shadow 0..31)
    const int nn = lane & 31;''','''    // node of its chain's tree of outcomes: nn = 0 is the state itself (lanes 0 and 32), 1..31 the nodes
    const int nn = l5;''')
sub('const int anc4 = anc * 4;','const int anc4 = (hh * 32 + anc) * 4;')
sub('const int lgo = (FL0 + 3 * F_LOGU) * 16 + (levc - 1) * 8;','const int lgo = hh * 1024 + (FL0 + 3 * F_LOGU) * 16 + (levc - 1) * 8;')
sub('const int tko = (FL0 + 3 * F_TEMP) * 16 + (levc - 1) * 8;','const int tko = hh * 1024 + (FL0 + 3 * F_TEMP) * 16 + (levc - 1) * 8;')
sub('''    const bool fl = lane < R * D;
    const int fu = fl ? lane / D : 0, fp = fl ? lane % D : 0;
    const int zao = ((fu * 2) * HW) * 16 + fp * 8;                      // second row: + HW * 16
    const int zto = (FL0 + 3 * fp) * 16 + fu * 8;
    const int ixown = (FL0 + 3 * F_IXOWN) * 16 + fu * 8;                // this pass's row indices (LIVE re-reads)''','''    const bool fl = l5 < R * D;
    const int fu = fl ? l5 / D : 0, fp = fl ? l5 % D : 0;
    const int zao = hh * 1024 + ((fu * 2) * HW) * 16 + fp * 8;          // second row: + HW * 16  (its own chain's KiB of the slot)
    const int zto = hh * 1024 + (FL0 + 3 * fp) * 16 + fu * 8;
    const int ixown = hh * 1024 + (FL0 + 3 * F_IXOWN) * 16 + fu * 8;    // this pass's row indices (LIVE re-reads)''')
# DMA offsets per chain
sub('''    unsigned int dma_off, dma_inc;
    {
        const unsigned int rec_off = (unsigned int)(reinterpret_cast<const unsigned char*>(P.rec_in) - zbase);
        if (rowl) { dma_off = (unsigned int)rj * 16u; dma_inc = 0u; }
        else if (fieldl && ff != F_TEMP) {
            const int rf = (ff == F_IXOWN || ff == F_IXNEXT) ? D + 1 : ff;
            dma_off = rec_off + (unsigned int)((((int64_t)rf * P.N + c) * P.rec_stride + (ff == F_IXNEXT ? PS2_AHEAD * R : 0)) * 8) + (unsigned int)fj * 16u;
            dma_inc = (unsigned int)(R * 8);
        } else if (fieldl) {
            dma_off = (unsigned int)(reinterpret_cast<const unsigned char*>(P.temperature) - zbase) + (unsigned int)fj * 16u;
            dma_inc = (unsigned int)(R * 8);
        } else { dma_off = 0u; dma_inc = 0u; }
    }''','''    // (every one of a pass's NCH DMA instructions uses all 64 lanes in these roles, for the wave's k-th chain)
    int64_t ck[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) ck[k] = (wv * NCH + k < P.N) ? wv * NCH + k : P.N - 1;
    unsigned int dma_off[NCH], dma_inc;
    {
        const unsigned int rec_off = (unsigned int)(reinterpret_cast<const unsigned char*>(P.rec_in) - zbase);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (rowl) { dma_off[k] = (unsigned int)rj * 16u; }
            else if (fieldl && ff != F_TEMP) {
                const int rf = (ff == F_IXOWN || ff == F_IXNEXT) ? D + 1 : ff;
                dma_off[k] = rec_off + (unsigned int)((((int64_t)rf * P.N + ck[k]) * P.rec_stride + (ff == F_IXNEXT ? PS2_AHEAD * R : 0)) * 8) + (unsigned int)fj * 16u;
            } else if (fieldl) {
                dma_off[k] = (unsigned int)(reinterpret_cast<const unsigned char*>(P.temperature) - zbase) + (unsigned int)fj * 16u;
            } else { dma_off[k] = 0u; }
        }
        dma_inc = fieldl ? (unsigned int)(R * 8) : 0u;
    }''')
# history
sub('''    const bool hl = lane < R * (D + 1);
    const int hj = hl ? lane / (D + 1) : 0, hp = hl ? lane % (D + 1) : 0;''','''    const bool hl = l5 < R * (D + 1) && act;
    const int hj = hl ? l5 / (D + 1) : 0, hp = hl ? l5 % (D + 1) : 0;''')
sub('const double* const tab_h = ct_w + hp;                 // + winner row * CR','const double* const tab_h = ct_w + hh * 32 * CR + hp;   // + winner row (of its own chain\'s tree) * CR')
# state
sub('''    if (P.safe_X) {                        // the state this launch starts from, kept for a redo (WindowParams::safe_X)
        double xv = x[0];
#pragma unroll
        for (int p = 1; p < D; ++p) xv = (lane == p) ? x[p] : xv;
        if (lane < D) P.safe_X[c + P.N * lane] = xv;
        if (lane == 0) P.safe_lp[c] = xlp;
    }
    if (lane < DP + 2) sd_w[R * DP + lane] = (lane < DP) ? -0.0 : 0.0;''','''    if (P.safe_X && act) {                 // the state this launch starts from, kept for a redo (WindowParams::safe_X)
        double xv = x[0];
#pragma unroll
        for (int p = 1; p < D; ++p) xv = (l5 == p) ? x[p] : xv;
        if (l5 < D) P.safe_X[c + P.N * l5] = xv;
        if (l5 == 0) P.safe_lp[c] = xlp;
    }
    if (l5 < DP + 2) sd_w[R * DP + l5] = (l5 < DP) ? -0.0 : 0.0;''')
sub('''    const double* rec_ix = P.rec_in + ((int64_t)(D + 1) * P.N + c) * P.rec_stride;
    uint64_t pp[PS2_AHEAD];
#pragma unroll
    for (int k = 0; k < PS2_AHEAD; ++k) pp[k] = (uint64_t)__double_as_longlong(rec_ix[k * R + ru]);''','''    uint64_t pp[NCH][PS2_AHEAD];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const double* rec_ix = P.rec_in + ((int64_t)(D + 1) * P.N + ck[q]) * P.rec_stride;
#pragma unroll
        for (int k = 0; k < PS2_AHEAD; ++k) pp[q][k] = (uint64_t)__double_as_longlong(rec_ix[k * R + ru]);
    }''')
sub('''#pragma unroll
    for (int k = 0; k < PS2_AHEAD; ++k) asm volatile("" :: "v"(pp[k]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");''','''#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int k = 0; k < PS2_AHEAD; ++k) asm volatile("" :: "v"(pp[q][k]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");''')
sub('''    auto issue = [&](uint64_t pack, int slot) __attribute__((always_inline)) {
        const unsigned int sel = __builtin_amdgcn_perm((unsigned int)(pack >> 32), (unsigned int)pack, selv);
        const unsigned int off = (sel << ZSH) + dma_off;
        dma_off += dma_inc;
#ifndef PS2_EXP_NODMA
#if PS2_DMA_BUFFER
        ps2_dma16b(zrsrc, off, raw_lds + (unsigned)slot * 1024u);
#else
        ps2_dma16(zbase, off, raw_lds + (unsigned)slot * 1024u);
#endif
#else
        asm volatile("" :: "v"(off));
#endif
    };
#pragma unroll
    for (int k = 0; k < PS2_AHEAD; ++k) issue(pp[k], k);''','''    // the DMAs of one pass: one instruction per chain of the wave, into that chain's KiB of the slot
    auto issue = [&](const uint64_t (&pack)[NCH], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const unsigned int sel = __builtin_amdgcn_perm((unsigned int)(pack[q] >> 32), (unsigned int)pack[q], selv);
            const unsigned int off = (sel << ZSH) + dma_off[q];
            dma_off[q] += dma_inc;
#if PS2_DMA_BUFFER
            ps2_dma16b(zrsrc, off, raw_lds + (unsigned)(slot * NCH + q) * 1024u);
#else
            ps2_dma16(zbase, off, raw_lds + (unsigned)(slot * NCH + q) * 1024u);
#endif
        }
    };
#pragma unroll
    for (int k = 0; k < PS2_AHEAD; ++k) {
        const uint64_t pk[NCH] = {pp[0][k], pp[1][k]};
        issue(pk, k);
    }''')
sub('uint64_t pr_f = 0;','uint64_t pr_f[NCH] = {0, 0};')
sub('''        const unsigned char* rw = raw_w + slot * 1024;
        // behind this slot's DMA in program order: AHEAD - 1 whole passes (a DMA each; !HRING: and a history store) and, !HRING,
        // this pass's store
#ifndef PS2_EXP_NOWAIT        // (timing experiments only -- results are garbage: scripts/ab_ps2.sh)
        if (counted) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HRING ? PS2_AHEAD - 1 : 2 * PS2_AHEAD - 1) : "memory");
#endif''','''        const unsigned char* rw = raw_w + slot * (NCH * 1024);
        // behind this slot's last DMA in program order: AHEAD - 1 whole passes (NCH DMAs each; !HRING: and a history store) and,
        // !HRING, this pass's store
        if (counted) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HRING ? NCH * (PS2_AHEAD - 1) : (1 + NCH) * (PS2_AHEAD - 1) + 1) : "memory");''')
sub('        pr_f = *reinterpret_cast<const uint64_t*>(rw + ixnext);','''#pragma unroll
        for (int q = 0; q < NCH; ++q) pr_f[q] = *reinterpret_cast<const uint64_t*>(rw + q * 1024 + ixnext);''')
sub('const uint64_t ix = *reinterpret_cast<const uint64_t*>(raw_w + slot * 1024 + ixown);','const uint64_t ix = *reinterpret_cast<const uint64_t*>(raw_w + slot * (NCH * 1024) + ixown);')
# pass body
sub('''            m32 = (unsigned int)__builtin_amdgcn_ballot_w64(logu_c < dlt);
            chg_a = __builtin_amdgcn_fcmp(d0, 0.0, 14 /* UNE */);
            chg_r = __builtin_amdgcn_fcmp(lpb - lpb, 0.0, 14);
            // on the path actually taken: every ancestor decided the way that leads here
            const bool onp = ((m32 ^ need1) & needm) == 0u;
            path = (unsigned int)__builtin_amdgcn_ballot_w64(onp) & 0xfffffffeu;
            accp = path & m32;''','''            m64 = __builtin_amdgcn_ballot_w64(logu_c < dlt);
            m32 = hh ? (unsigned int)(m64 >> 32) : (unsigned int)m64;           // its own chain's accept mask
            chg_a = __builtin_amdgcn_fcmp(d0, 0.0, 14 /* UNE */);
            chg_r = __builtin_amdgcn_fcmp(lpb - lpb, 0.0, 14);
            // on the path actually taken: every ancestor decided the way that leads here
            const bool onp = ((m32 ^ need1) & needm) == 0u;
            path64 = __builtin_amdgcn_ballot_w64(onp) & 0xfffffffefffffffeull;
            path = hh ? (unsigned int)(path64 >> 32) : (unsigned int)path64;
            accp = path & m32;''')
sub('''        unsigned int m32, path, accp;
        unsigned long long chg_a, chg_r;''','''        unsigned int m32, path, accp;
        unsigned long long chg_a, chg_r, m64, path64;''')
sub('            const double* wr = ct_w + win * CR;','            const double* wr = ct_w + (hh * 32 + win) * CR;')
sub('''            const unsigned int chm = (accp & (unsigned int)chg_a) | (path & ~m32 & (unsigned int)chg_r);
            cnt_total += (unsigned int)__builtin_popcount(chm);
            if constexpr (FIRST) cnt_first = (chm >> 1) & 1u;''','''            const unsigned long long chm = ((path64 & m64 & chg_a) | (path64 & ~m64 & chg_r)) & actm;
            cnt_total += (unsigned int)__builtin_popcountll(chm);
            if constexpr (FIRST) cnt_first = (unsigned int)((chm >> 1) & 1ull) + (unsigned int)((chm >> 33) & 1ull);''')
sub('''            const double v = ct_w[win * CR + ((lane < D) ? lane : 0)];''','''            const double v = ct_w[(hh * 32 + win) * CR + ((l5 < D) ? l5 : 0)];''')
sub('''                if (lane < D) pub_rows[(w * PS_PUB + (int)((unsigned int)nb % PS_PUB)) * D + lane] = v;''','''                if (l5 < D) pub_rows[((w * PS_PUB + (int)((unsigned int)nb % PS_PUB)) * NCH + hh) * D + l5] = v;''')
sub('''                if (lane < D && P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + lane] = v;''','''                if (l5 < D && act && P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + l5] = v;''')
sub('''            if (lane < D && P.snap) P.snap[nb * P.N * D + c + P.N * lane] = v;''','''            if (l5 < D && act && P.snap) P.snap[nb * P.N * D + c + P.N * l5] = v;''')
sub('''        double xv = x[0];
#pragma unroll
        for (int p = 1; p < D; ++p) xv = (lane == p) ? x[p] : xv;
        if (lane < D) P.Xcur[c + P.N * lane] = xv;
        if (lane == 0) P.lpcur[c] = xlp;
    }
    wave_store_counts(P, c, cnt_total, cnt_first);''','''        double xv = x[0];
#pragma unroll
        for (int p = 1; p < D; ++p) xv = (l5 == p) ? x[p] : xv;
        if (l5 < D && act) P.Xcur[c + P.N * l5] = xv;
        if (l5 == 0 && act) P.lpcur[c] = xlp;
    }
    wave_store_counts(P, wv, cnt_total, cnt_first);''')
sub('if (P.stamps && lane == 0 && c < 65536) {','if (P.stamps && lane == 0 && wv < 65536) {')
sub('unsigned long long* o = P.stamps + (size_t)c * 16;','unsigned long long* o = P.stamps + (size_t)wv * 16;')
# pc_produce in !LIVE: unchanged.  PS2_EXP_NODMA removed above.
hdr='''// demcz_kernels_ps2d.h -- window_kernel_ps2 (demcz_kernels_ps2.h) with TWO chains to a wave (round 4).
//
// A pass of the steady-state wave-per-chain consumer uses 31 of a wave's 64 lanes for the tree of outcomes; lanes 32..63 only
// shadowed lanes 0..31.  Here they run a second chain: lane l works for chain NCH * wave + (l >> 5), as node l & 31 of THAT
// chain's tree.  What a wave has once per chain: a KiB of every raw slot and a DMA instruction per pass (a pass's rows, normals,
// log u and indices are 54 of an instruction's 64 lanes), a block of increments, half of the candidate table (rows 0..31 /
// 32..63), a row in the publisher's ring.  What it has once: the pass structure, the boundary counter, the history hand-off (30 + 30
// of its lanes), the accept compare -- one ballot whose halves are the two chains' masks -- and the instruction stream: the pass
// costs what it cost (one DMA instruction more) and resolves five generations of TWO chains.  The LIVE launch therefore holds
// 2048 chains where it held 1024 (one five-wave workgroup of eight chains per CU), at the same time per launch.
// An odd population: the last wave's second half shadows the last chain (reads what it reads, writes nothing, counts nothing).
// Same arithmetic on the same values as window_kernel_ps2, hence the oracle's bits.
#pragma once

#include "demcz_kernels_ps2.h"

#pragma clang fp contract(off)

namespace demcz {

'''
out = hdr + body + '}  // namespace demcz\n'
target = CSRC / "demcz_kernels_ps2d.h"
if len(sys.argv) > 1 and sys.argv[1] == "--check":
    sys.exit(0 if target.read_text() == out else 1)
target.write_text(out)
print("written")
