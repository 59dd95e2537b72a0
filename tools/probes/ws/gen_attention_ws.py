#!/usr/bin/env python
"""Generator of the warp-specialised attention kernel body for gfx950 (successor of gen_attention_r64.py).

Writes self-forcing_amd/csrc/attention_ws_asm.inc: ONE inline-asm string, the whole kernel after the C++
prologue of `attention_ws_kernel` in attention.hip.  Run it again after editing; the .inc is committed.

Why: the 64-rows-per-wave kernel (gen_attention_r64.py) halves the LDS traffic per flop but needs all 512
registers of a SIMD lane for ONE wave, and in a single in-order stream every non-MFMA instruction (softmax
VALU, LDS fragment reads, LDS-DMA requests) adds its issue time to the MFMA time: 958 us where the MFMAs
alone need 572.  Here the same 64 query rows are handled by TWO waves that share a SIMD:

  role A (waves 0-3)  S^T = K Q^T + (-m c) for both 32-query blocks, lazy-rescale online softmax, row sums;
                      writes the bf16 P^T fragments (already in MFMA B-operand layout) to LDS
  role B (waves 4-7)  one tile behind: reads the P^T fragments, O^T += V^T P^T; issues all LDS-DMA requests;
                      applies the (rare) rescale factors A leaves in LDS; normalises and stores O

Each role fits in half a SIMD's registers (160 VGPR + 96 AGPR), so the SIMD's issue port interleaves A's VALU
with B's MFMAs and vice versa, while every K / V^T fragment is still read once per 64 query rows.
One s_barrier per tile; K ring of 3 slots, V ring of 2, P ring of 2 tiles; 156 KB of LDS.

Step t (after barrier t):   A: softmax of units 2t, 2t+1 of tile t, QK of units 2t+1, 2t+2  -> P(t) into LDS
                            B: O += V(t-1) P(t-1); requests K(t+2), V(t)
"""
import os
import sys

ABL = set()          # timing-only ablations: nosoftmax, nodma, nobarrier, nopv (role B without MFMAs), noqk
if "--abl" in sys.argv:
    ABL = set(sys.argv[sys.argv.index("--abl") + 1].split(","))

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "self-forcing_amd", "csrc",
                   "attention_ws_asm.inc")

# ---- inline-asm operands (inputs only)
(K_SRD, V_SRD, TILE_BYTES, NTILES, LK, CSCALE, LDS_WAVE, TID4, LDS_BASE, ROLE) = ("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "%8", "%9")

N_PARAM = 32
PARAM_STRIDE = 2048          # 512 threads x 4 bytes
# LDS map (bytes from the base): K ring 3 x 16 KiB | V ring 2 x 16 KiB | P ring | hand-off area
V_BASE = 3 * 16384
P_BASE = 5 * 16384           # 80 KiB; + parity*32768 + pair*8192 + unit*4096 + frag*1024 + lane*16
H_BASE = P_BASE + 65536      # + parity*6144 + pair*1536 + unit*768 + {D0: 0, D1: 256, FLAG: 512} + lane*4
L_BASE = H_BASE + 12288       # 1 / l hand-off: + pair*512 + qb*256 + lane*4
LDS_BYTES = L_BASE + 2048

# ---- registers.  hipcc splits a 256-register budget evenly when AGPRs are used (128 VGPR + 128 AGPR), so
#      role A (scores + softmax) keeps Q^T and the K fragments in AGPRs and packs the bf16 P^T fragments IN PLACE
#      into the low registers of each score block; role B keeps O^T in AGPRs, its V^T / P^T fragments in VGPRs.
A_KADDR = [1 + i for i in range(8)]      # advanced in place from ring slot to ring slot
A_QP = [10, 12]                           # prologue only; afterwards scratch
A_HH4, A_PADDR, A_HADDR, A_LADDR = 9, 14, 15, 16
A_T = [17, 18, 19, 20, 10, 11, 12, 13]    # scratch registers ([1],[2] form an even-aligned pair)
A_MX = [21, 22]
A_MRC = [23, 24]
A_L2 = [26, 28]
A_S = [32, 64]         # A_S[buf] + 16 qb; after the convert the P fragment (qb, ks) is A_S[buf] + 16 qb + 4 ks .. +4
A_MINIT = [96, 112]
A_Q = lambda qb, s: (qb * 8 + s) * 4      # noqa: E731   AGPR a0..a63
A_KF = 64                                   # AGPR a64..a95, 8 slots

B_VLO = [1 + i for i in range(4)]        # V / P / hand-off addresses are advanced in place by +-slot strides
B_VHI = [5 + i for i in range(4)]
B_DMAOFF = [9 + i for i in range(4)]
B_LADDR = 13
B_OP = [14, 16]
B_VALID = 18           # bit qb: query block qb has rows inside Lq
B_PADDR, B_HADDR = 20, 21
B_TMP = 22             # v22..v31 scratch
B_VF = lambda b: 32 + b * 32   # noqa: E731   VGPR, double buffered by unit
B_PF = lambda b: 96 + b * 16   # noqa: E731
B_O = lambda qb, db: (qb * 4 + db) * 16   # noqa: E731   AGPR a0..a127
ZTMP = 96              # v96..v101: scratch of the LDS zeroing in both prologues

# ---- SGPRs (asm-owned)
ST, NTM1, KS_CUR, KS_N1, KS_DMA, VS_CUR, VS_DMA, STMP = 60, 61, 62, 63, 64, 65, 66, 67
C2, SEXEC, SOFFK, SOFFV, SLIM, STMP2, SPAR = 68, 70, 72, 73, 74, 75, 76
SVD, SPD, SHD = 77, 78, 79     # role B: signed strides to the other V slot / P parity / hand-off parity
THR = "0x41000000"    # 8.0

out = []
cold = []
label_n = [0]


def e(s):
    out.append(s)


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=1):
    return f"a{base}" if n == 1 else f"a[{base}:{base + n - 1}]"


def new_label(stem):
    label_n[0] += 1
    return f".Lws_{stem}_{label_n[0]}%="


class Lds:
    """in-order LDS-op tracker -> exact s_waitcnt lgkmcnt before each consumer"""

    def __init__(self):
        self.issued = 0
        self.done = 0

    def op(self, text):
        e(text)
        self.issued += 1
        return self.issued - 1

    def need(self, idx):
        if idx < self.done or idx < 0:
            return
        after = self.issued - 1 - idx
        e(f"s_waitcnt lgkmcnt({min(after, 15)})")
        if after <= 15:
            self.done = max(self.done, idx + 1)

    def reset(self, in_flight):
        self.issued, self.done = in_flight, 0


lds = Lds()
tags = {}


def mix(a, b):
    if not a:
        return list(b)
    res, bi = [], 0
    for i, x in enumerate(a):
        res.append(x)
        want = len(b) * (i + 1) // len(a)
        while bi < want:
            res.append(b[bi])
            bi += 1
    return res + list(b[bi:])


def emit_item(item):
    if item[0] == "lds":
        tags[item[2]] = lds.op(item[1])
    else:
        e(item[1])


def phase(mfmas, others):
    """emit `mfmas` with `others` spread evenly behind them; LDS waits are derived from the tracker"""
    if not mfmas:
        for item in others:
            emit_item(item)
        return
    n = len(mfmas)
    per = [len(others) * (i + 1) // n - len(others) * i // n for i in range(n)]
    oi = 0
    for i, (text, needs) in enumerate(mfmas):
        for tg in needs:
            lds.need(tags[tg])
        if text:
            e(text)
        for _ in range(per[i]):
            emit_item(others[oi])
            oi += 1


# ======================================================================================== role A
def a_k_reads(kb):
    return [("lds", f"ds_read_b128 {ar(A_KF + 4 * s, 4)}, {vr(A_KADDR[s])} offset:{kb * 8192}", ("k", s)) for s in range(8)]


def a_qk_mfmas(sbuf, zero_init=False):
    m = []
    for s in range(8):
        for qb in range(2):
            acc = vr(A_S[sbuf] + 16 * qb, 16)
            src_c = acc if s else ("0" if zero_init else vr(A_MINIT[qb], 16))
            text = f"v_mfma_f32_32x32x16_bf16 {acc}, {ar(A_KF + 4 * s, 4)}, {ar(A_Q(qb, s), 4)}, {src_c}"
            m.append(("" if "noqk" in ABL and not zero_init else text, [("k", s)]))
    return m


def a_max_items(sbuf, masked, kb):
    it = []
    sreg = lambda qb, r: A_S[sbuf] + 16 * qb + r   # noqa: E731
    if "nosoftmax" in ABL:
        return it
    if masked:
        it.append(("valu", f"v_mov_b32 {vr(A_T[7])}, 0xf149f2ca"))
        for r in range(16):
            it.append(("valu", f"s_sub_i32 s{STMP}, s{SLIM}, {32 * kb + (r & 3) + 8 * (r >> 2)}"))
            it.append(("valu", f"v_cmp_le_i32 vcc, s{STMP}, {vr(A_HH4)}"))
            for qb in range(2):
                it.append(("valu", f"v_cndmask_b32 {vr(sreg(qb, r))}, {vr(sreg(qb, r))}, {vr(A_T[7])}, vcc"))
    for qb in range(2):
        it.append(("valu", f"v_max3_f32 {vr(A_MX[qb])}, {vr(sreg(qb, 0))}, {vr(sreg(qb, 1))}, {vr(sreg(qb, 2))}"))
    for j in range(6):
        for qb in range(2):
            it.append(("valu", f"v_max3_f32 {vr(A_MX[qb])}, {vr(A_MX[qb])}, {vr(sreg(qb, 3 + 2 * j))}, {vr(sreg(qb, 4 + 2 * j))}"))
    for qb in range(2):
        it.append(("valu", f"v_max_f32 {vr(A_MX[qb])}, {vr(A_MX[qb])}, {vr(sreg(qb, 15))}"))
    for qb in range(2):
        it.append(("valu", f"v_mov_b32 {vr(A_T[qb])}, {vr(A_MX[qb])}"))
    for qb in range(2):
        it.append(("valu", f"s_nop 0\n\tv_permlane32_swap_b32 {vr(A_MX[qb])}, {vr(A_T[qb])}"))
    for qb in range(2):
        it.append(("valu", f"s_nop 0\n\tv_max_f32 {vr(A_MX[qb])}, {vr(A_MX[qb])}, {vr(A_T[qb])}"))
    return it


def a_rescale_check(sbuf, unit):
    """Lazy rescale decision of one unit, and the hand-off to role B: FLAG (0 / 1) and, when set, the per-row
    amounts d (B multiplies O by 2^-d before it accumulates this unit).  A's own state moves too."""
    h_off = unit * 768
    if "nosoftmax" in ABL:
        e(f"v_mov_b32 {vr(A_T[2])}, 0")
        tags[("flagw", unit)] = lds.op(f"ds_write_b32 {vr(A_HADDR)}, {vr(A_T[2])} offset:{h_off + 512}")
        return
    blk, back = new_label("rescale"), new_label("rescaled")
    e(f"v_max_f32 {vr(A_T[2])}, {vr(A_MX[0])}, {vr(A_MX[1])}")
    e(f"v_cmp_lt_f32 vcc, {THR}, {vr(A_T[2])}")
    e(f"v_mov_b32 {vr(A_T[3])}, 0")
    e(f"s_cbranch_vccnz {blk}")
    e(f"{back}:")
    tags[("flagw", unit)] = lds.op(f"ds_write_b32 {vr(A_HADDR)}, {vr(A_T[3])} offset:{h_off + 512}")
    c = [f"{blk}:"]
    for qb in range(2):
        d = A_T[4 + qb]
        c.append(f"v_max_f32 {vr(d)}, 0, {vr(A_MX[qb])}")
        c.append(f"v_add_f32 {vr(A_MRC[qb])}, {vr(A_MRC[qb])}, {vr(d)}")
        c.append(f"v_exp_f32 {vr(A_T[2])}, -{vr(d)}")
        c.append("s_nop 1")
        c.append(f"v_mul_f32 {vr(A_L2[qb])}, {vr(A_L2[qb])}, {vr(A_T[2])}")
        c.append(f"v_mul_f32 {vr(A_L2[qb] + 1)}, {vr(A_L2[qb] + 1)}, {vr(A_T[2])}")
        for r in range(16):
            c.append(f"v_sub_f32 {vr(A_S[sbuf] + 16 * qb + r)}, {vr(A_S[sbuf] + 16 * qb + r)}, {vr(d)}")
        for r in range(16):
            c.append(f"v_sub_f32 {vr(A_MINIT[qb] + r)}, {vr(A_MINIT[qb] + r)}, {vr(d)}")
        c.append(f"ds_write_b32 {vr(A_HADDR)}, {vr(d)} offset:{h_off + 256 * qb}")
    c += [f"v_mov_b32 {vr(A_T[3])}, 1", "s_waitcnt lgkmcnt(0)", "s_nop 4", f"s_branch {back}"]
    cold.extend(c)


def a_exp_items(sbuf):
    """p = exp2(s c - m c) in place; row sums (the values must survive: plain accumulation, no in-place tree);
    then the bf16 P^T fragments packed in place: pair (r, r+1) of block qb -> register r/2 of the block"""
    it = []
    sreg = lambda qb, r: A_S[sbuf] + 16 * qb + r   # noqa: E731
    if "nosoftmax" not in ABL:
        for r in range(16):
            for qb in range(2):
                it.append(("valu", f"v_exp_f32 {vr(sreg(qb, r))}, {vr(sreg(qb, r))}"))
        # row sums with plain v_add_f32 (packed fp32 instructions do not execute beside MFMAs); the p values must
        # survive for the convert, so two running sums per block instead of an in-place tree
        for r in range(0, 16, 2):
            for qb in range(2):
                it.append(("valu", f"v_add_f32 {vr(A_L2[qb])}, {vr(A_L2[qb])}, {vr(sreg(qb, r))}"))
                it.append(("valu", f"v_add_f32 {vr(A_L2[qb] + 1)}, {vr(A_L2[qb] + 1)}, {vr(sreg(qb, r + 1))}"))
    for r in range(0, 16, 2):
        for qb in range(2):
            it.append(("valu", f"v_cvt_pk_bf16_f32 {vr(sreg(qb, r >> 1))}, {vr(sreg(qb, r))}, {vr(sreg(qb, r + 1))}"))
    return it


def a_p_writes(sbuf, unit):
    """the four P^T fragments of a unit (f = 2 qb + ks), lane-linear: lane l's 16 bytes at +16 l"""
    return [("lds", f"ds_write_b128 {vr(A_PADDR)}, {vr(A_S[sbuf] + 16 * (f >> 1) + 4 * (f & 1), 4)} offset:{unit * 4096 + f * 1024}", ("pw", unit, f))
            for f in range(4)]


def a_advance_k():
    """K fragment addresses: from ring slot KS_CUR to KS_N1, in place"""
    r = [("valu", f"s_sub_u32 s{STMP}, s{KS_N1}, s{KS_CUR}")]
    return r + [("valu", f"v_add_u32 {vr(A_KADDR[s])}, s{STMP}, {vr(A_KADDR[s])}") for s in range(8)]


def a_phase(mfmas, valu, kreads):
    """16 QK MFMAs; the VALU / LDS-write items spread evenly behind them; the K fragment of head-dim step s of
    the NEXT unit is read into slot s right after the two MFMAs that consume slot s (write-after-read safe)"""
    if not mfmas:
        for item in valu:
            emit_item(item)
        return
    n = len(mfmas)
    per = [len(valu) * (i + 1) // n - len(valu) * i // n for i in range(n)]
    vi = 0
    for i, (text, needs) in enumerate(mfmas):
        for tg in needs:
            lds.need(tags[tg])
        if text:
            e(text)
        if i % 2 == 1 and kreads:
            emit_item(kreads[i // 2])
        for _ in range(per[i]):
            emit_item(valu[vi])
            vi += 1


def a_step(kind):
    """role A, one tile: kind 'normal' | 'penult' | 'last'"""
    last = kind == "last"
    lds.reset(8)
    tags.clear()
    tags.update({("k", s): s for s in range(8)})
    if "nobarrier" not in ABL:
        e("s_barrier")
    if kind == "penult":
        e(f"s_add_u32 s{STMP}, s{ST}, 1")
        e(f"s_lshl_b32 s{STMP}, s{STMP}, 6")
        e(f"s_sub_i32 s{SLIM}, {LK}, s{STMP}")
    if last:
        e(f"s_lshl_b32 s{STMP}, s{ST}, 6")
        e(f"s_sub_i32 s{SLIM}, {LK}, s{STMP}")
    # unit 0: its scores are in S0 (previous QK), maxima in MX
    a_rescale_check(0, 0)
    if not last:
        for _, t in a_advance_k():          # K fragment addresses of tile t+1 (the reads in flight already left)
            e(t)
    e("s_nop 1")
    a_phase(a_qk_mfmas(1), a_exp_items(0) + [("valu", "s_nop 0")] + a_p_writes(0, 0), [] if last else a_k_reads(0))
    e("s_nop 7")
    for _, t in a_max_items(1, last, 1):
        e(t)
    a_rescale_check(1, 1)
    e("s_nop 2")
    a_phase([] if last else a_qk_mfmas(0), a_exp_items(1) + [("valu", "s_nop 0")] + a_p_writes(1, 1), [] if last else a_k_reads(1))
    if not last:
        e("s_nop 7")
        for _, t in a_max_items(0, kind == "penult", 0):
            e(t)
        # K ring (cur, n1, dma) <- (n1, dma, cur); P parity toggles
        e(f"s_mov_b32 s{STMP}, s{KS_CUR}")
        e(f"s_mov_b32 s{KS_CUR}, s{KS_N1}")
        e(f"s_mov_b32 s{KS_N1}, s{KS_DMA}")
        e(f"s_mov_b32 s{KS_DMA}, s{STMP}")
        e(f"v_add_u32 {vr(A_PADDR)}, s{SPD}, {vr(A_PADDR)}")     # P / hand-off parity alternates
        e(f"v_add_u32 {vr(A_HADDR)}, s{SHD}, {vr(A_HADDR)}")
        e(f"s_sub_i32 s{SPD}, 0, s{SPD}")
        e(f"s_sub_i32 s{SHD}, 0, s{SHD}")
        e(f"s_add_u32 s{ST}, s{ST}, 1")
    lds.need(tags[("pw", 1, 3)])          # this tile's P / flag writes are in LDS before the next barrier


def role_a():
    e("; ================= role A: scores, softmax, P -> LDS")
    dst = {0: A_KADDR, 20: [A_QP[0], A_QP[0] + 1, A_QP[1], A_QP[1] + 1], 29: [A_PADDR, A_HADDR, A_LADDR]}
    for base, regs in dst.items():
        for i, r in enumerate(regs):
            e(f"ds_read_b32 {vr(r)}, {TID4} offset:{(base + i) * PARAM_STRIDE}")
    # 4 * (lane >> 5): which half of a 32-key block's rows this lane's accumulator registers hold
    e(f"v_subrev_u32 {vr(A_HH4)}, {LDS_BASE}, {TID4}")
    e(f"v_lshrrev_b32 {vr(A_HH4)}, 7, {vr(A_HH4)}")       # (tid*4) >> 7 = tid >> 5
    e(f"v_and_b32 {vr(A_HH4)}, 1, {vr(A_HH4)}")
    e(f"v_lshlrev_b32 {vr(A_HH4)}, 2, {vr(A_HH4)}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                      # 1: parameters read
    zero_lds()
    e("s_barrier")                      # 2: LDS zeroed
    e(f"s_mov_b32 s{C2}, {CSCALE}")
    e(f"s_mov_b32 s{C2 + 1}, {CSCALE}")
    e(f"s_mov_b32 s{SPD}, 32768")
    e(f"s_mov_b32 s{SHD}, 6144")
    for qb in range(2):
        e(f"v_mov_b32 {vr(A_L2[qb])}, 0")
        e(f"v_mov_b32 {vr(A_L2[qb] + 1)}, 0")
    for qb in range(2):
        for sidx in range(8):
            e(f"global_load_dwordx4 {vr(A_S[0] + (qb * 8 + sidx) * 4, 4)}, {vr(A_QP[qb], 2)}, off offset:{sidx * 32}")
    e("s_waitcnt vmcnt(0)")
    t0, t1 = A_T[1], A_T[2]             # even-aligned pair
    for i in range(64):
        x = A_S[0] + i
        e(f"v_lshlrev_b32 {vr(t0)}, 16, {vr(x)}")
        e(f"v_and_b32 {vr(t1)}, 0xffff0000, {vr(x)}")
        e(f"v_pk_mul_f32 {vr(t0, 2)}, {vr(t0, 2)}, s[{C2}:{C2 + 1}]")
        e(f"v_cvt_pk_bf16_f32 {vr(x)}, {vr(t0)}, {vr(t1)}")
        e(f"v_accvgpr_write_b32 {ar(i)}, {vr(x)}")
    e("s_nop 4")
    e("s_barrier")                      # 3: K(0), K(1) landed (requested by role B)
    lds.reset(0)
    tags.clear()
    a_phase([], a_k_reads(0), [])
    a_phase(a_qk_mfmas(0, zero_init=True), [], [])
    a_phase([], a_k_reads(1), [])
    e("s_nop 7")
    e("s_nop 7")
    e("s_nop 7")
    e(f"s_mov_b32 s{SLIM}, {LK}")
    for _, t in a_max_items(0, True, 0):
        e(t)
    for qb in range(2):
        if "nosoftmax" in ABL:
            e(f"v_mov_b32 {vr(A_MX[qb])}, 0")
        e(f"v_mov_b32 {vr(A_MRC[qb])}, {vr(A_MX[qb])}")
        for r in range(16):
            e(f"v_sub_f32 {vr(A_MINIT[qb] + r)}, 0, {vr(A_MX[qb])}")
        for r in range(16):
            e(f"v_sub_f32 {vr(A_S[0] + 16 * qb + r)}, {vr(A_S[0] + 16 * qb + r)}, {vr(A_MX[qb])}")
        e(f"v_mov_b32 {vr(A_MX[qb])}, 0")
    e("s_nop 4")
    loop, last_l, pen_l = ".Lws_a_loop%=", ".Lws_a_last%=", ".Lws_a_penult%="
    e(f"{loop}:")
    e(f"s_cmp_eq_u32 s{ST}, s{NTM1}")
    e(f"s_cbranch_scc1 {last_l}")
    e(f"s_add_u32 s{STMP}, s{ST}, 1")
    e(f"s_cmp_eq_u32 s{STMP}, s{NTM1}")
    e(f"s_cbranch_scc1 {pen_l}")
    a_step("normal")
    e(f"s_branch {loop}")
    e(f"{pen_l}:")
    a_step("penult")
    e(f"{last_l}:")
    a_step("last")
    # 1 / l for role B
    for qb in range(2):
        e(f"v_add_f32 {vr(A_L2[qb])}, {vr(A_L2[qb])}, {vr(A_L2[qb] + 1)}")
        e(f"v_mov_b32 {vr(A_T[0])}, {vr(A_L2[qb])}")
        e("s_nop 0")
        e(f"v_permlane32_swap_b32 {vr(A_L2[qb])}, {vr(A_T[0])}")
        e("s_nop 0")
        e(f"v_add_f32 {vr(A_L2[qb])}, {vr(A_L2[qb])}, {vr(A_T[0])}")
        e(f"v_rcp_f32 {vr(A_T[1])}, {vr(A_L2[qb])}")
        e("s_nop 1")
        e(f"ds_write_b32 {vr(A_LADDR)}, {vr(A_T[1])} offset:{256 * qb}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                      # step nt: role B's last tile
    e("s_branch .Lws_end%=")


# ======================================================================================== role B
def b_v_reads(kb, buf):
    r = []
    for ks in range(2):
        for db in range(4):
            f = ks * 4 + db
            off = (2 * kb + ks) * 4096
            r.append(("lds", f"ds_read_b64_tr_b16 {vr(B_VF(buf) + 4 * f, 2)}, {vr(B_VLO[db])} offset:{off}", ("vlo", buf, f)))
            r.append(("lds", f"ds_read_b64_tr_b16 {vr(B_VF(buf) + 4 * f + 2, 2)}, {vr(B_VHI[db])} offset:{off}", ("vhi", buf, f)))
    return r


def b_p_reads(unit):
    r = [("lds", f"ds_read_b32 {vr(B_TMP + unit)}, {vr(B_HADDR)} offset:{unit * 768 + 512}", ("flag", unit))]
    r += [("lds", f"ds_read_b128 {vr(B_PF(unit) + 4 * f, 4)}, {vr(B_PADDR)} offset:{unit * 4096 + f * 1024}", ("p", unit, f)) for f in range(4)]
    return r


def b_pv_mfmas(buf):
    m = []
    for ks in range(2):
        for db in range(4):
            f = ks * 4 + db
            for qb in range(2):
                acc = ar(B_O(qb, db), 16)
                text = f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(B_VF(buf) + 4 * f, 4)}, {vr(B_PF(buf) + 4 * (2 * qb + ks), 4)}, {acc}"
                m.append(("" if "nopv" in ABL else text, [("vlo", buf, f), ("vhi", buf, f), ("p", buf, 2 * qb + ks)]))
    return m


def b_rescale_check(unit):
    """apply role A's rescale of this unit (rare): O *= 2^-d per query column"""
    blk, back = new_label("brescale"), new_label("brescaled")
    lds.need(tags[("flag", unit)])
    e(f"v_cmp_ne_u32 vcc, 0, {vr(B_TMP + unit)}")
    e("s_nop 1")
    e(f"s_cbranch_vccnz {blk}")
    e(f"{back}:")
    c = [f"{blk}:"] + ["s_nop 7"] * 12      # the P.V MFMAs issued just before may still be writing O
    for qb in range(2):
        c.append(f"ds_read_b32 {vr(B_TMP + 2)}, {vr(B_HADDR)} offset:{unit * 768 + 256 * qb}")
        c.append("s_waitcnt lgkmcnt(0)")
        c.append(f"v_exp_f32 {vr(B_TMP + 2)}, -{vr(B_TMP + 2)}")
        c.append("s_nop 7")
        c.append("s_nop 7")
        for base in range(B_O(qb, 0), B_O(qb, 0) + 64, 6):
            n = min(6, B_O(qb, 0) + 64 - base)
            for i in range(n):
                c.append(f"v_accvgpr_read_b32 {vr(B_TMP + 3 + i)}, {ar(base + i)}")
            for i in range(n):
                c.append(f"v_mul_f32 {vr(B_TMP + 3 + i)}, {vr(B_TMP + 3 + i)}, {vr(B_TMP + 2)}")
            for i in range(n):
                c.append(f"v_accvgpr_write_b32 {ar(base + i)}, {vr(B_TMP + 3 + i)}")
    c += ["s_nop 4", f"s_branch {back}"]
    cold.extend(c)


def b_dma_items(srd, soff_sreg, slot_sreg):
    if "nodma" in ABL:
        return []
    it = []
    for i in range(4):
        txt = [f"s_add_u32 s{STMP2}, s{slot_sreg}, {LDS_WAVE}", f"s_add_u32 m0, s{STMP2}, {i * 1024}", "s_nop 1",
               f"buffer_load_dwordx4 {vr(B_DMAOFF[i])}, {srd}, s{soff_sreg} offen lds"]
        it.append(("raw", "\n\t".join(txt)))
    return it


def clamp_tile(dst_sreg, ahead):
    e(f"s_add_u32 s{STMP}, s{ST}, {ahead}")
    e(f"s_min_u32 s{STMP}, s{STMP}, s{NTM1}")
    e(f"s_mul_i32 s{dst_sreg}, s{STMP}, {TILE_BYTES}")


def b_step(kind):
    """role B at step t: kind 'first' (t = 0: requests only), 'normal', 'final' (t = nt: last tile, no requests)"""
    lds.reset(0)
    tags.clear()
    e("s_waitcnt vmcnt(0)")
    if "nobarrier" not in ABL:
        e("s_barrier")
    dmas = []
    if kind != "final":
        clamp_tile(SOFFK, 2)
        clamp_tile(SOFFV, 0)
        dmas = b_dma_items(V_SRD, SOFFV, VS_DMA) + b_dma_items(K_SRD, SOFFK, KS_DMA)
    if kind == "first":
        for _, t in dmas:
            e(t)
    else:
        # tile t-1 (V slot / P parity / hand-off parity are where the address registers point)
        # LDS-DMA requests between the MFMAs of the first unit, one per two MFMAs (tools/probes/dma_probe.hip: a
        # burst of 8 blocks the wave for ~220 ns, the same 8 between 16 MFMAs cost ~20 ns); the second unit's
        # MFMAs are the time they have to land before the next barrier
        phase([], b_p_reads(0) + b_v_reads(0, 0))
        b_rescale_check(0)
        phase(b_pv_mfmas(0), mix(dmas, b_p_reads(1) + b_v_reads(1, 1)))
        b_rescale_check(1)
        phase(b_pv_mfmas(1), [])
    if kind != "final":
        # K ring (cur, n1, dma) <- (n1, dma, cur); V ring (cur, dma) swap
        e(f"s_mov_b32 s{STMP}, s{KS_CUR}")
        e(f"s_mov_b32 s{KS_CUR}, s{KS_N1}")
        e(f"s_mov_b32 s{KS_N1}, s{KS_DMA}")
        e(f"s_mov_b32 s{KS_DMA}, s{STMP}")
        e(f"s_mov_b32 s{STMP}, s{VS_CUR}")
        e(f"s_mov_b32 s{VS_CUR}, s{VS_DMA}")
        e(f"s_mov_b32 s{VS_DMA}, s{STMP}")
        if kind != "first":      # a tile was consumed: its V slot, P parity and hand-off parity alternate
            for d in range(4):
                e(f"v_add_u32 {vr(B_VLO[d])}, s{SVD}, {vr(B_VLO[d])}")
                e(f"v_add_u32 {vr(B_VHI[d])}, s{SVD}, {vr(B_VHI[d])}")
            e(f"v_add_u32 {vr(B_PADDR)}, s{SPD}, {vr(B_PADDR)}")
            e(f"v_add_u32 {vr(B_HADDR)}, s{SHD}, {vr(B_HADDR)}")
            for sr in (SVD, SPD, SHD):
                e(f"s_sub_i32 s{sr}, 0, s{sr}")
        e(f"s_add_u32 s{ST}, s{ST}, 1")


def role_b():
    e("; ================= role B: P.V, LDS-DMA requests, output")
    dst = {8: B_VLO + B_VHI + B_DMAOFF, 24: [B_OP[0], B_OP[0] + 1, B_OP[1], B_OP[1] + 1], 28: [B_VALID, B_PADDR, B_HADDR, B_LADDR]}
    for base, regs in dst.items():
        for i, r in enumerate(regs):
            e(f"ds_read_b32 {vr(r)}, {TID4} offset:{(base + i) * PARAM_STRIDE}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                      # 1
    zero_lds()
    e("s_barrier")                      # 2
    e(f"s_mov_b32 s{SOFFK}, 0")
    for _, t in b_dma_items(K_SRD, SOFFK, KS_CUR):
        e(t)
    e(f"s_min_u32 s{STMP}, s{NTM1}, 1")
    e(f"s_mul_i32 s{SOFFK}, s{STMP}, {TILE_BYTES}")
    for _, t in b_dma_items(K_SRD, SOFFK, KS_N1):
        e(t)
    for i in range(128):
        e(f"v_accvgpr_write_b32 {ar(i)}, 0")
    for d in range(4):                  # V fragment addresses: slot 0 of the V ring first
        e(f"v_add_u32 {vr(B_VLO[d])}, {V_BASE}, {vr(B_VLO[d])}")
        e(f"v_add_u32 {vr(B_VHI[d])}, {V_BASE}, {vr(B_VHI[d])}")
    e(f"s_mov_b32 s{SVD}, 16384")
    e(f"s_mov_b32 s{SPD}, 32768")
    e(f"s_mov_b32 s{SHD}, 6144")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")                      # 3: K(0), K(1) in LDS
    b_step("first")
    loop, fin = ".Lws_b_loop%=", ".Lws_b_final%="
    e(f"{loop}:")
    e(f"s_cmp_gt_u32 s{ST}, s{NTM1}")     # t == nt ?
    e(f"s_cbranch_scc1 {fin}")
    b_step("normal")
    e(f"s_branch {loop}")
    e(f"{fin}:")
    b_step("final")
    # normalise with role A's 1 / l and store (the fragment registers are free now)
    for _ in range(12):
        e("s_nop 7")
    w = B_VF(0)
    for qb in range(2):
        e(f"ds_read_b32 {vr(B_TMP)}, {vr(B_LADDR)} offset:{256 * qb}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_and_b32 {vr(B_TMP + 1)}, {1 << qb}, {vr(B_VALID)}")
        e(f"v_cmp_ne_u32 vcc, 0, {vr(B_TMP + 1)}")
        e(f"s_and_saveexec_b64 s[{SEXEC}:{SEXEC + 1}], vcc")
        for db in range(4):
            for i in range(16):
                e(f"v_accvgpr_read_b32 {vr(w + i)}, {ar(B_O(qb, db) + i)}")
            e("s_nop 1")
            for i in range(16):
                e(f"v_mul_f32 {vr(w + i)}, {vr(w + i)}, {vr(B_TMP)}")
            for i in range(8):
                e(f"v_cvt_pk_bf16_f32 {vr(w + 16 + i)}, {vr(w + 2 * i)}, {vr(w + 2 * i + 1)}")
            for rg in range(4):
                e(f"global_store_dwordx2 {vr(B_OP[qb], 2)}, {vr(w + 16 + 2 * rg, 2)}, off offset:{db * 64 + rg * 16}")
            e("s_waitcnt vmcnt(0)")
        e(f"s_mov_b64 exec, s[{SEXEC}:{SEXEC + 1}]")
    e("s_branch .Lws_end%=")


def zero_lds():
    """K and V slots only (80 KiB): rows past Lk are never fetched; 0 * stale NaN would poison P.V"""
    for i in range(4):
        e(f"v_mov_b32 {vr(ZTMP + i)}, 0")
    e(f"v_subrev_u32 {vr(ZTMP + 4)}, {LDS_BASE}, {TID4}")
    e(f"v_lshlrev_b32 {vr(ZTMP + 4)}, 2, {vr(ZTMP + 4)}")
    e(f"v_add_u32 {vr(ZTMP + 4)}, {LDS_BASE}, {vr(ZTMP + 4)}")        # lds base + tid*16 (512 threads: 8 KiB per round)
    e(f"v_add_u32 {vr(ZTMP + 5)}, 0x10000, {vr(ZTMP + 4)}")
    for i in range(10):
        e(f"ds_write_b128 {vr(ZTMP + 4 + (i // 8))}, {vr(ZTMP, 4)} offset:{(i % 8) * 8192}")
    e("s_waitcnt lgkmcnt(0)")


def main():
    e(f"s_sub_u32 s{NTM1}, {NTILES}, 1")
    e(f"s_mov_b32 s{ST}, 0")
    e(f"s_mov_b32 s{SPAR}, 0")
    for i, sr in enumerate((KS_CUR, KS_N1, KS_DMA)):
        e(f"s_mov_b32 s{sr}, {i * 16384}")
    e(f"s_mov_b32 s{VS_CUR}, {V_BASE + 16384}")      # step 0 requests V(0) into VS_DMA = slot 0; the swap makes it VS_CUR at step 1
    e(f"s_mov_b32 s{VS_DMA}, {V_BASE}")
    e(f"s_cmp_eq_u32 {ROLE}, 0")
    e("s_cbranch_scc0 .Lws_role_b%=")
    role_a()
    e(".Lws_role_b%=:")
    role_b()
    for ln in cold:
        e(ln)
    e(".Lws_end%=:")
    body_txt = "\n".join('    "' + ln.replace("\n\t", '\\n\\t') + '\\n"' for ln in out)
    clob_v = ", ".join(f'"v{i}"' for i in range(1, 128))
    clob_a = ", ".join(f'"a{i}"' for i in range(0, 128))
    clob_s = ", ".join(f'"s{i}"' for i in range(60, 80))
    path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_attention_ws.py -- do not edit; see that file for the design.\n")
        f.write(f"#define SF_WS_N_PARAM {N_PARAM}\n#define SF_WS_LDS_BYTES {LDS_BYTES}\n#define SF_WS_P_BASE {P_BASE}\n#define SF_WS_H_BASE {H_BASE}\n")
        f.write("#define SF_WS_ASM_BODY \\\n" + body_txt.replace("\n", " \\\n") + "\n")
        f.write(f"#define SF_WS_CLOBBERS {clob_v}, {clob_a}, {clob_s}, \"vcc\", \"scc\", \"memory\"\n")
    print(f"wrote {path}: {len(out)} asm lines")


if __name__ == "__main__":
    main()
