#!/usr/bin/env python
"""Generator of the hand-scheduled 64-query-rows-per-wave attention kernel body for gfx950.

Writes self-forcing_amd/csrc/attention_r64_asm.inc: ONE inline-asm string, the whole kernel after the
C++ prologue of `attention_r64_kernel` in attention.hip (which computes the per-lane addresses with the
same formulas as the other attention kernels and hands them over through LDS).  Run it again after
editing; the .inc is committed so that the build needs no Python.

Why assembly: with 64 query rows per wave every K / V^T fragment read from LDS feeds TWO MFMAs (19 instead
of 34 LDS bytes per kflop -- the measured limiter of the 32-row kernels), but the wave then needs all 512
registers (one wave per SIMD) and its softmax VALU, LDS reads, LDS-DMA requests and MFMAs share ONE
in-order instruction stream: the matrix pipe stays busy only if no more than ~28 cycles of other
instructions sit between two MFMAs (tools/probes/overlap_probe.hip), which hipcc's scheduler does not
arrange.  Everything here is about taking instructions out of that stream and spreading the rest evenly:

  * the query block is pre-scaled by c = log2(e)/sqrt(d), and the QK^T accumulation STARTS from -m c (a
    16-register tuple per query block, rewritten only on a rescale) instead of 0: the MFMA delivers
    s c - m c directly, the softmax is exp2 + convert, no multiply / subtract pass;
  * lazy rescale: m is only a reference point, so O and l are rescaled only when a row's maximum exceeds
    it by more than 8 (in log2 units; P <= 2^8).  With O in AGPRs a rescale costs ~400 instructions; the
    exact policy fired in most tiles (any of 128 rows seeing a new maximum) and cost 40 % of the kernel;
  * row maxima and row sums run under the P.V MFMAs, exp2 + convert under the QK^T MFMAs;
  * K / V^T fragments are read from LDS straight into AGPRs (MFMA A operands), Q^T lives in AGPRs too.

Per wave: 64 queries = two 32-query blocks (MFMA 32x32x16 column blocks).  Keys in tiles of 64 = two
32-key units.  Registers:
  AGPR  a[0:127] O^T accumulators O[qb][db] | a[128:191] Q^T fragments Q[qb][s] | a[192:223] K fragments
        | a[224:255] V^T fragments
  VGPR  S0, S1 (2 x 32) scores of the unit being produced / consumed; P (16) bf16 P^T fragments;
        MINIT (2 x 16) = -m c broadcast; state m c, l (pairs), row maxima
Pipeline per tile t (units u0 = 2t, u1 = 2t+1); one s_barrier per tile; K ring of 4 slots, V ring of 3:
  0  s_waitcnt vmcnt(4), barrier                      (everything but the latest V request has landed)
  A  [rescale check u0]  QK(u1) -> S1  ||  exp2 / convert of S0 -> P  ||  read V^T fragments of u0
  B  PV(u0)  ||  row sums of S0, row maxima of S1  ||  read K fragments of unit 0 of tile t+1  ||  request K(t+3)
  C  [rescale check u1]  QK(2t+2) -> S0  ||  exp2 / convert of S1 -> P  ||  read V^T fragments of u1
  D  PV(u1)  ||  row sums of S1, row maxima of S0  ||  read K fragments of unit 1 of tile t+1  ||  request V(t+2)
The last tile runs with the key mask (keys >= Lk -> -1e30) and without successor work; the tile before it
takes the maxima of the last tile's first unit with the mask.
"""
import os
import sys

ABL = set()          # timing-only ablations: nosoftmax, nolds, nomfma, nodma, nobarrier
if "--abl" in sys.argv:
    ABL = set(sys.argv[sys.argv.index("--abl") + 1].split(","))

ALIGN = int(sys.argv[sys.argv.index("--align") + 1]) if "--align" in sys.argv else 6    # log2 bytes; 0 = none
PAD = int(sys.argv[sys.argv.index("--pad") + 1]) if "--pad" in sys.argv else 0          # extra 4-byte s_nops behind it
# how many of a wave's LDS-DMA requests may still be in flight behind the wait at the top of a (steady-state) tile
TOP_VMCNT = int(sys.argv[sys.argv.index("--top-vmcnt") + 1]) if "--top-vmcnt" in sys.argv else 4
# where a tile's eight LDS-DMA requests are issued: K pieces placed in phases A / B, V pieces in C / D ("n_in_A,n_in_C":
# how many of the four go to the FIRST phase of each pair; 0,0 = all four K requests in B and all four V requests in D)
DMA_SPLIT = [int(x) for x in (sys.argv[sys.argv.index("--dma-split") + 1] if "--dma-split" in sys.argv else "0,0").split(",")]

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "self-forcing_amd", "csrc",
                   "attention_r64_asm.inc")

# ---- inline-asm operands (inputs only)
K_SRD, V_SRD, TILE_BYTES, NTILES, LK, CSCALE, LDS_WAVE, TID4, LDS_BASE = "%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "%8"

# ---- register map
KADDR = [1 + i for i in range(8)]
VLO = [9 + i for i in range(4)]
VHI = [13 + i for i in range(4)]
DMAOFF = [17 + i for i in range(4)]
QP = [22, 24]
OP = [26, 28]
HH4, VALID = 30, [31, 32]
KCUR = [33 + i for i in range(8)]
VCURLO = [41 + i for i in range(4)]
VCURHI = [45 + i for i in range(4)]
S = [64, 96]          # S[buf] + qb*16
P = 128               # P + qb*8 + ks*4
MINIT = [144, 160]
TMP = 176             # 24 scratch registers
MX = [240, 241]
MRC = [242, 243]      # m c per query block (log2 units)
L2 = [244, 246]
DLT = [248, 250]      # rescale amount
NEG = 254
A_O = lambda qb, db: (qb * 4 + db) * 16          # noqa: E731
A_Q = lambda qb, s: 128 + (qb * 8 + s) * 4       # noqa: E731
KF, VF = 192, 224     # AGPR fragment slots
N_PARAM = 31

ST, NTM1, KS_CUR, KS_N1, KS_N2, KS_DMA, VS_CUR, VS_N1, VS_DMA, STMP = 60, 61, 62, 63, 64, 65, 66, 67, 68, 69
C2, SEXEC, SOFFK, SOFFV, SLIM, STMP2 = 70, 72, 74, 75, 76, 77
V_BASE = 4 * 16384
THR = "0x41000000"    # 8.0

out = []


def e(s):
    out.append(s)


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=1):
    return f"a{base}" if n == 1 else f"a[{base}:{base + n - 1}]"


class Lds:
    """in-order LDS read tracker -> exact s_waitcnt lgkmcnt before each consumer"""

    def __init__(self):
        self.issued = 0
        self.done = 0      # reads with index < done are known complete

    def read(self, text):
        e(text)
        self.issued += 1
        return self.issued - 1

    def need(self, idx):
        if idx < self.done or idx < 0:
            return
        after = self.issued - 1 - idx
        if "nowait" not in ABL:
            e(f"s_waitcnt lgkmcnt({min(after, 15)})")
        if after <= 15:
            self.done = max(self.done, idx + 1)

    def reset(self, in_flight):
        self.issued, self.done = in_flight, 0


lds = Lds()


def k_reads(kb):
    """the 8 K fragments (one per 16-wide head-dim step) of a 32-key unit -> AGPR slots"""
    return [("lds", f"ds_read_b128 {ar(KF + 4 * s, 4)}, {vr(KCUR[s])} offset:{kb * 8192}", ("k", s)) for s in range(8)]


def v_reads(kb):
    """the 8 V^T fragments (2 key steps x 4 head-dim blocks) of a 32-key unit -> AGPR slots (lo, hi halves)"""
    r = []
    for ks in range(2):
        for db in range(4):
            f = ks * 4 + db
            off = (2 * kb + ks) * 4096
            r.append(("lds", f"ds_read_b64_tr_b16 {ar(VF + 4 * f, 2)}, {vr(VCURLO[db])} offset:{off}", ("vlo", f)))
            r.append(("lds", f"ds_read_b64_tr_b16 {ar(VF + 4 * f + 2, 2)}, {vr(VCURHI[db])} offset:{off}", ("vhi", f)))
    return r


def qk_mfmas(sbuf, zero_init=False):
    """S^T = K Q^T + (-m c): the accumulation starts from the MINIT tuple"""
    m = []
    for s in range(8):
        for qb in range(2):
            acc = vr(S[sbuf] + 16 * qb, 16)
            src_c = acc if s else ("0" if zero_init else vr(MINIT[qb], 16))
            m.append((f"v_mfma_f32_32x32x16_bf16 {acc}, {ar(KF + 4 * s, 4)}, {ar(A_Q(qb, s), 4)}, {src_c}", [("k", s)]))
    return m


def pv_mfmas():
    m = []
    for ks in range(2):
        for db in range(4):
            f = ks * 4 + db
            for qb in range(2):
                acc = ar(A_O(qb, db), 16)
                m.append((f"v_mfma_f32_32x32x16_bf16 {acc}, {ar(VF + 4 * f, 4)}, {vr(P + 8 * qb + 4 * ks, 4)}, {acc}", [("vlo", f), ("vhi", f)]))
    return m


label_n = [0]
cold = []     # out-of-line blocks (the rare rescale), emitted behind the kernel's last instruction


def new_label(stem):
    label_n[0] += 1
    return f".Lr64_{stem}_{label_n[0]}%="


def max_items(sbuf, masked, kb):
    """row maxima of one unit's scores (already s c - m c) for both query blocks -> MX"""
    it = []
    sreg = lambda qb, r: S[sbuf] + 16 * qb + r   # noqa: E731
    if "nosoftmax" in ABL or "nomax" in ABL:
        return it
    if masked:
        # key = 64 t + 32 kb + (r&3) + 8 (r>>2) + 4 hh ; SLIM = Lk - 64 t
        for r in range(16):
            it.append(("valu", f"s_sub_i32 s{STMP}, s{SLIM}, {32 * kb + (r & 3) + 8 * (r >> 2)}"))
            it.append(("valu", f"v_cmp_le_i32 vcc, s{STMP}, {vr(HH4)}"))
            for qb in range(2):
                it.append(("valu", f"v_cndmask_b32 {vr(sreg(qb, r))}, {vr(sreg(qb, r))}, {vr(NEG)}, vcc"))
    for qb in range(2):
        it.append(("valu", f"v_max3_f32 {vr(MX[qb])}, {vr(sreg(qb, 0))}, {vr(sreg(qb, 1))}, {vr(sreg(qb, 2))}"))
    for j in range(6):
        for qb in range(2):
            it.append(("valu", f"v_max3_f32 {vr(MX[qb])}, {vr(MX[qb])}, {vr(sreg(qb, 3 + 2 * j))}, {vr(sreg(qb, 4 + 2 * j))}"))
    for qb in range(2):
        it.append(("valu", f"v_max_f32 {vr(MX[qb])}, {vr(MX[qb])}, {vr(sreg(qb, 15))}"))
    for qb in range(2):   # across the lane pair (l, l+32) that shares a query column
        it.append(("valu", f"v_mov_b32 {vr(TMP + qb)}, {vr(MX[qb])}"))
    # hazard: a VGPR written by a VALU instruction needs two wait states before v_permlane32_swap reads it (hipcc puts
    # `s_nop 1` between a v_mov and the swap); the swap's results need none.  With the two query blocks interleaved --
    # mov0, mov1, s_nop 0, swap0, swap1, max0, max1 -- ONE s_nop would cover both; `--abl fewnops` drops the other three
    # (bit-identical, and within 0.2 % in time at every Lk, round 3: these slots sit in the P.V phases, which are not the
    # ones that bound the tile), so the stream keeps the conservative four.
    old = "fewnops" not in ABL
    for qb in range(2):
        pre = "s_nop 0\n\t" if (old or qb == 0) else ""
        it.append(("valu", f"{pre}v_permlane32_swap_b32 {vr(MX[qb])}, {vr(TMP + qb)}"))
    for qb in range(2):
        pre = "s_nop 0\n\t" if old else ""
        it.append(("valu", f"{pre}v_max_f32 {vr(MX[qb])}, {vr(MX[qb])}, {vr(TMP + qb)}"))
    return it


def rescale_check(sbuf):
    """emitted in front of a QK phase: if some row of this unit exceeds its reference by more than 2^8,
    move the reference: m c += d, l *= 2^-d, O *= 2^-d, this unit's scores -= d, MINIT -= d  (d = max(mx, 0))"""
    if "nosoftmax" in ABL:
        return
    blk, back = new_label("rescale"), new_label("rescaled")
    e(f"v_max_f32 {vr(TMP + 2)}, {vr(MX[0])}, {vr(MX[1])}")
    e(f"v_cmp_lt_f32 vcc, {THR}, {vr(TMP + 2)}")
    e("s_nop 1")
    e(f"s_cbranch_vccnz {blk}")      # rare: the common path falls through
    e(f"{back}:")
    c = [f"{blk}:"]
    for qb in range(2):
        d = DLT[qb]
        c.append(f"v_max_f32 {vr(d)}, 0, {vr(MX[qb])}")
        c.append(f"v_add_f32 {vr(MRC[qb])}, {vr(MRC[qb])}, {vr(d)}")
        c.append(f"v_exp_f32 {vr(TMP + 2)}, -{vr(d)}")
        c.append("s_nop 1")
        c.append(f"v_mov_b32 {vr(TMP + 3)}, {vr(TMP + 2)}")
        c.append(f"v_pk_mul_f32 {vr(L2[qb], 2)}, {vr(L2[qb], 2)}, {vr(TMP + 2, 2)}")
        for base in range(A_O(qb, 0), A_O(qb, 0) + 64, 8):
            for i in range(8):
                c.append(f"v_accvgpr_read_b32 {vr(TMP + 4 + i)}, {ar(base + i)}")
            for i in range(0, 8, 2):
                c.append(f"v_pk_mul_f32 {vr(TMP + 4 + i, 2)}, {vr(TMP + 4 + i, 2)}, {vr(TMP + 2, 2)}")
            for i in range(8):
                c.append(f"v_accvgpr_write_b32 {ar(base + i)}, {vr(TMP + 4 + i)}")
        for r in range(16):
            c.append(f"v_sub_f32 {vr(S[sbuf] + 16 * qb + r)}, {vr(S[sbuf] + 16 * qb + r)}, {vr(d)}")
        for r in range(16):
            c.append(f"v_sub_f32 {vr(MINIT[qb] + r)}, {vr(MINIT[qb] + r)}, {vr(d)}")
    c += ["s_nop 4", f"s_branch {back}"]
    cold.extend(c)


def exp_items(sbuf):
    """p = exp2(s c - m c) in place, then the bf16 P^T fragments"""
    it = []
    sreg = lambda qb, r: S[sbuf] + 16 * qb + r   # noqa: E731
    if "nosoftmax" not in ABL and "noexp" not in ABL:
        for r in range(16):
            for qb in range(2):
                it.append(("valu", f"v_exp_f32 {vr(sreg(qb, r))}, {vr(sreg(qb, r))}"))
    for r in range(0, 16, 2):   # P^T fragment of key step ks = r >> 3, element pair (r & 7) >> 1
        for qb in range(2):
            dst = P + 8 * qb + 4 * (r >> 3) + ((r & 7) >> 1)
            it.append(("valu", f"v_cvt_pk_bf16_f32 {vr(dst)}, {vr(sreg(qb, r))}, {vr(sreg(qb, r + 1))}"))
    it.append(("valu", "s_nop 1"))
    return it


def sum_items(sbuf):
    """row sums of the p values left in S[sbuf] (done under the P.V MFMAs).  Plain v_add_f32, pairwise tree in
    place (the p values are dead once converted): packed fp32 instructions (v_pk_add_f32) do NOT execute
    beside MFMAs -- 48 of them between 8 MFMAs cost 139 ns on top of the MFMAs' 130 ns, 48 v_add_f32 cost 3 ns
    (tools/probes/overlap_probe.hip) -- so the 16 packed adds of the first version cost 65 us per launch.  Nor does
    v_dot2c_f32_bf16 acc, P_pair, (1.0, 1.0) on the bf16 P fragments (32 instead of 64 instructions per tile): measured
    1.5-2.5 % SLOWER than this tree at Lk = 32760."""
    if "nosoftmax" in ABL or "nosum" in ABL:
        return []
    it = []
    sr = lambda qb, r: vr(S[sbuf] + 16 * qb + r)   # noqa: E731
    for step in (1, 2, 4, 8):
        for r in range(0, 16, 2 * step):
            for qb in range(2):
                it.append(("valu", f"v_add_f32 {sr(qb, r)}, {sr(qb, r)}, {sr(qb, r + step)}"))
    for qb in range(2):
        it.append(("valu", f"v_add_f32 {vr(L2[qb])}, {vr(L2[qb])}, {sr(qb, 0)}"))
    return it


def mix(a, b):
    """b spread evenly through a (order within each list kept)"""
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


def phase(mfmas, others):
    """emit `mfmas` with `others` spread evenly behind them; LDS waits are derived from the tracker"""
    tags = phase.tags
    if "nomfma" in ABL:
        mfmas = []
    for flag, kinds in (("nolds", ("k", "vlo", "vhi")), ("novread", ("vlo", "vhi")), ("nokread", ("k",))):
        if flag in ABL:
            for it in others:
                if it[0] == "lds" and it[2][0] in kinds:
                    tags[it[2]] = -1
            others = [it for it in others if not (it[0] == "lds" and it[2][0] in kinds)]
    n = max(1, len(mfmas))
    per = [len(others) * (i + 1) // n - len(others) * i // n for i in range(n)] if mfmas else []
    oi = 0

    def emit_other(item):
        if item[0] == "lds":
            tags[item[2]] = lds.read(item[1])
        else:
            e(item[1])

    if not mfmas:
        for item in others:
            emit_other(item)
        return
    for i, (text, needs) in enumerate(mfmas):
        for tg in needs:
            lds.need(tags[tg])
        if "mfma16" in ABL:    # timing only: the same flops as two 16x16x32 instructions (numerically meaningless)
            import re
            mm = re.match(r"v_mfma_f32_32x32x16_bf16 ([av])\[(\d+):\d+\], (\S+), (\S+), (\S+)$", text)
            kind, lo, a_op, b_op, c_op = mm.group(1), int(mm.group(2)), mm.group(3), mm.group(4), mm.group(5)
            for h in range(2):
                acc = f"{kind}[{lo + 4 * h}:{lo + 4 * h + 3}]"
                if c_op == "0":
                    c = "0"
                else:
                    cm = re.match(r"([av])\[(\d+):\d+\]", c_op)
                    c = f"{cm.group(1)}[{int(cm.group(2)) + 4 * h}:{int(cm.group(2)) + 4 * h + 3}]"
                e(f"v_mfma_f32_16x16x32_bf16 {acc}, {a_op}, {b_op}, {c}")
        else:
            e(text)
        for _ in range(per[i]):
            emit_other(others[oi])
            oi += 1
    while oi < len(others):
        emit_other(others[oi])
        oi += 1


phase.tags = {}


def dma_items(srd, soff_sreg, slot_sreg, prologue=False):
    """this wave's four 1 KiB pieces of a K or V tile: global -> LDS, range-checked, swizzle on the source"""
    if "nodma" in ABL and not prologue:
        return []
    it = []
    for i in range(4):
        # ONE M0 per group of four pieces: piece i's LDS address is M0 + inst_offset (i KiB) + 16 lane, and the same
        # inst_offset is taken back out of the piece's global byte offset (DMAOFF[i], prepared by the C++ prologue), so the
        # global address and its range check are unchanged.  (M0 per piece was two SALU + a wait state each: 8 pieces a tile.)
        txt = [f"buffer_load_dwordx4 {vr(DMAOFF[i])}, {srd}, s{soff_sreg} offen" + (f" offset:{i * 1024}" if i else "") + " lds"]
        if i == 0:
            txt = [f"s_add_u32 m0, s{slot_sreg}, {LDS_WAVE}", "s_nop 1"] + txt
        it.append(("raw", "\n\t".join(txt)))
    return it


def dma(srd, soff_sreg, slot_sreg, prologue=False):
    for _, t in dma_items(srd, soff_sreg, slot_sreg, prologue):
        e(t)


def set_kcur(slot_sreg):
    if "noaddr" in ABL:      # timing-only: what the per-tile ring addressing costs (upper bound of unrolling the rings)
        return []
    return [("valu", f"v_add_u32 {vr(KCUR[s])}, s{slot_sreg}, {vr(KADDR[s])}") for s in range(8)]


def set_vcur(slot_sreg):
    if "noaddr" in ABL:
        return []
    return ([("valu", f"v_add_u32 {vr(VCURLO[d])}, s{slot_sreg}, {vr(VLO[d])}") for d in range(4)]
            + [("valu", f"v_add_u32 {vr(VCURHI[d])}, s{slot_sreg}, {vr(VHI[d])}") for d in range(4)])


def clamp_tile(dst_sreg, ahead):
    e(f"s_add_u32 s{STMP}, s{ST}, {ahead}")
    e(f"s_min_u32 s{STMP}, s{STMP}, s{NTM1}")
    e(f"s_mul_i32 s{dst_sreg}, s{STMP}, {TILE_BYTES}")


def body(kind):
    """one key tile; kind: 'normal', 'penult' (the next tile is the last one: its first unit's maxima are
    taken with the key mask) or 'last' (masked, no successor)"""
    last = kind == "last"
    # in flight at entry: the 8 K fragment reads of unit 1 of tile t (issued by the previous phase D / prologue);
    # MX holds the row maxima of unit 0 (taken in the previous phase D / prologue)
    lds.reset(8)
    phase.tags = {("k", s): s for s in range(8)}
    e(f"s_waitcnt vmcnt({TOP_VMCNT})" if kind == "normal" else "s_waitcnt vmcnt(0)")
    if "nobarrier" not in ABL:
        e("s_barrier")
    dma_k = dma_v = []
    if not last:
        clamp_tile(SOFFK, 3)
        clamp_tile(SOFFV, 2)
        dma_k, dma_v = dma_items(K_SRD, SOFFK, KS_DMA), dma_items(V_SRD, SOFFV, VS_DMA)
    if kind == "penult":
        e(f"s_add_u32 s{STMP}, s{ST}, 1")
        e(f"s_lshl_b32 s{STMP}, s{STMP}, 6")
        e(f"s_sub_i32 s{SLIM}, {LK}, s{STMP}")       # Lk - 64 (t+1)
    if last:
        e(f"s_lshl_b32 s{STMP}, s{ST}, 6")
        e(f"s_sub_i32 s{SLIM}, {LK}, s{STMP}")
    # A: QK(u1) -> S1 || exp2 / convert of u0 (S0 -> P) || V^T fragments of u0
    rescale_check(0)
    ka, vc = DMA_SPLIT
    phase(qk_mfmas(1), mix(mix(exp_items(0), v_reads(0)), dma_k[:ka]))
    # B: PV(u0) || row sums of u0, row maxima of u1 || K fragments of unit 0 of tile t+1 || request K(t+3)
    pad = [("valu", "s_nop 7"), ("valu", "s_nop 7")] if last else []
    phase(pv_mfmas(), mix(sum_items(0) + ([] if last else set_kcur(KS_N1)) + pad + mix(max_items(1, last, 1), [] if last else k_reads(0)), dma_k[ka:]))
    # C: QK(2t+2) -> S0 || exp2 / convert of u1 (S1 -> P) || V^T fragments of u1
    rescale_check(1)
    phase([] if last else qk_mfmas(0), v_reads(1) + exp_items(1) if last else mix(mix(exp_items(1), v_reads(1)), dma_v[:vc]))
    # D: PV(u1) || row sums of u1, row maxima of unit 2t+2 || K fragments of unit 1 of tile t+1 || request V(t+2)
    phase(pv_mfmas(), mix(sum_items(1) + ([] if last else mix(max_items(0, kind == "penult", 0), k_reads(1))), dma_v[vc:]))
    if not last and "noaddr" in ABL:
        e(f"s_add_u32 s{ST}, s{ST}, 1")
    elif not last:
        # rotate the rings: K (cur, n1, n2, dma) <- (n1, n2, dma, cur); V (cur, n1, dma) <- (n1, dma, cur)
        e(f"s_mov_b32 s{STMP}, s{KS_CUR}")
        e(f"s_mov_b32 s{KS_CUR}, s{KS_N1}")
        e(f"s_mov_b32 s{KS_N1}, s{KS_N2}")
        e(f"s_mov_b32 s{KS_N2}, s{KS_DMA}")
        e(f"s_mov_b32 s{KS_DMA}, s{STMP}")
        e(f"s_mov_b32 s{STMP}, s{VS_CUR}")
        e(f"s_mov_b32 s{VS_CUR}, s{VS_N1}")
        e(f"s_mov_b32 s{VS_N1}, s{VS_DMA}")
        e(f"s_mov_b32 s{VS_DMA}, s{STMP}")
        for _, t in set_vcur(VS_CUR):
            e(t)
        e(f"s_add_u32 s{ST}, s{ST}, 1")


def main():
    e("; ---- parameters handed over through LDS: dword j of lane tid at byte j*1024 + tid*4")
    dst = KADDR + VLO + VHI + DMAOFF + [QP[0], QP[0] + 1, QP[1], QP[1] + 1, OP[0], OP[0] + 1, OP[1], OP[1] + 1, HH4] + VALID
    assert len(dst) == N_PARAM
    for j, d in enumerate(dst):
        e(f"ds_read_b32 {vr(d)}, {TID4} offset:{j * 1024}")
        if j % 8 == 7:
            e("s_waitcnt lgkmcnt(0)")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e("; ---- constants and state")
    e(f"s_mov_b32 s{C2}, {CSCALE}")
    e(f"s_mov_b32 s{C2 + 1}, {CSCALE}")
    e(f"v_mov_b32 {vr(NEG)}, 0xf149f2ca")          # -1e30
    for qb in range(2):
        e(f"v_mov_b32 {vr(L2[qb])}, 0")
        e(f"v_mov_b32 {vr(L2[qb] + 1)}, 0")
    e(f"s_sub_u32 s{NTM1}, {NTILES}, 1")
    e(f"s_mov_b32 s{ST}, 0")
    for i, sr in enumerate((KS_CUR, KS_N1, KS_N2, KS_DMA)):
        e(f"s_mov_b32 s{sr}, {i * 16384}")
    for i, sr in enumerate((VS_CUR, VS_N1, VS_DMA)):
        e(f"s_mov_b32 s{sr}, {V_BASE + i * 16384}")
    e("; ---- Q^T fragments: requested FIRST (they need no LDS), so that the LDS zeroing and the first K / V requests run under them")
    for qb in range(2):
        for s in range(8):
            e(f"global_load_dwordx4 {vr(S[0] + (qb * 8 + s) * 4, 4)}, {vr(QP[qb], 2)}, off offset:{s * 32}")
    e("; ---- zero the seven LDS slots (rows past Lk are never fetched; 0 * stale NaN would poison P.V)")
    for i in range(4):
        e(f"v_mov_b32 {vr(TMP + i)}, 0")
    e(f"v_subrev_u32 {vr(TMP + 4)}, {LDS_BASE}, {TID4}")    # TID4 = lds base + tid*4
    e(f"v_lshlrev_b32 {vr(TMP + 4)}, 2, {vr(TMP + 4)}")
    e(f"v_add_u32 {vr(TMP + 4)}, {LDS_BASE}, {vr(TMP + 4)}")   # lds base + tid*16
    e(f"v_add_u32 {vr(TMP + 5)}, 0x10000, {vr(TMP + 4)}")
    for i in range(28):
        e(f"ds_write_b128 {vr(TMP + 4 + i // 16)}, {vr(TMP, 4)} offset:{(i % 16) * 4096}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e("; ---- first tiles: K(0), V(0), K(1), V(1), K(2) (tile indices clamped to the last tile): 20 requests per wave, in this order")
    e(f"s_mov_b32 s{SOFFK}, 0")
    dma(K_SRD, SOFFK, KS_CUR, True)
    dma(V_SRD, SOFFK, VS_CUR, True)
    clamp_tile(SOFFK, 1)
    dma(K_SRD, SOFFK, KS_N1, True)
    dma(V_SRD, SOFFK, VS_N1, True)
    clamp_tile(SOFFK, 2)
    dma(K_SRD, SOFFK, KS_N2, True)
    e("; ---- O^T = 0; Q^T scaled by c -> AGPRs as soon as Q has landed (the 20 tile requests stay in flight behind it)")
    for i in range(128):
        e(f"v_accvgpr_write_b32 {ar(i)}, 0")
    e("s_waitcnt vmcnt(20)")
    for i in range(64):
        x = S[0] + i
        e(f"v_lshlrev_b32 {vr(TMP)}, 16, {vr(x)}")
        e(f"v_and_b32 {vr(TMP + 1)}, 0xffff0000, {vr(x)}")
        e(f"v_pk_mul_f32 {vr(TMP, 2)}, {vr(TMP, 2)}, s[{C2}:{C2 + 1}]")
        e(f"v_cvt_pk_bf16_f32 {vr(x)}, {vr(TMP)}, {vr(TMP + 1)}")
        e(f"v_accvgpr_write_b32 {ar(128 + i)}, {vr(x)}")
    e("s_nop 4")
    e("; ---- K(0) is all the first product needs: wait for this wave's four pieces of it, then for the other waves'")
    e("s_waitcnt vmcnt(16)")
    e("s_barrier")
    for _, t in set_kcur(KS_CUR):
        e(t)
    for _, t in set_vcur(VS_CUR):
        e(t)
    e("; ---- QK of unit 0 from zero, K fragments of unit 1 in flight for the loop, first reference m = row max")
    lds.reset(0)
    phase.tags = {}
    phase([], k_reads(0))
    phase(qk_mfmas(0, zero_init=True), [])
    phase([], k_reads(1))
    e("s_nop 7")
    e("s_nop 7")
    e("s_nop 7")
    e(f"s_mov_b32 s{SLIM}, {LK}")
    for _, t in max_items(0, True, 0):
        e(t)
    for qb in range(2):
        if "nosoftmax" in ABL:
            e(f"v_mov_b32 {vr(MX[qb])}, 0")
        e(f"v_mov_b32 {vr(MRC[qb])}, {vr(MX[qb])}")
        for r in range(16):
            e(f"v_sub_f32 {vr(MINIT[qb] + r)}, 0, {vr(MX[qb])}")
        for r in range(16):
            e(f"v_sub_f32 {vr(S[0] + 16 * qb + r)}, {vr(S[0] + 16 * qb + r)}, {vr(MX[qb])}")
        e(f"v_mov_b32 {vr(MX[qb])}, 0")
    e("s_nop 4")
    loop, last_l, pen_l = ".Lr64_loop%=", ".Lr64_last%=", ".Lr64_penult%="
    # the loop head's placement is pinned (a hand-written stream is sensitive to where it sits relative to the fetch
    # granule: MI355X_MICROARCH.md, 'Code-placement sensitivity'), so edits to the prologue do not move the loop body
    if ALIGN:
        e(f".p2align {ALIGN}")
        for _ in range(PAD):
            e("s_nop 0")
    e(f"{loop}:")
    e(f"s_cmp_eq_u32 s{ST}, s{NTM1}")
    e(f"s_cbranch_scc1 {last_l}")
    e(f"s_add_u32 s{STMP}, s{ST}, 1")
    e(f"s_cmp_eq_u32 s{STMP}, s{NTM1}")
    e(f"s_cbranch_scc1 {pen_l}")
    body("normal")
    e(f"s_branch {loop}")
    e(f"{pen_l}:")
    body("penult")
    e(f"{last_l}:")
    body("last")
    e("; ---- epilogue: O / l -> bf16, rows past Lq masked off")
    e("s_nop 7")
    e("s_nop 7")
    e("s_nop 7")
    for qb in range(2):
        e(f"v_add_f32 {vr(L2[qb])}, {vr(L2[qb])}, {vr(L2[qb] + 1)}")
        e(f"v_mov_b32 {vr(TMP)}, {vr(L2[qb])}")
        e("s_nop 0")
        e(f"v_permlane32_swap_b32 {vr(L2[qb])}, {vr(TMP)}")
        e("s_nop 0")
        e(f"v_add_f32 {vr(L2[qb])}, {vr(L2[qb])}, {vr(TMP)}")
        e(f"v_rcp_f32 {vr(DLT[0])}, {vr(L2[qb])}")
        e("s_nop 1")
        e(f"v_cmp_ne_u32 vcc, 0, {vr(VALID[qb])}")
        e(f"s_and_saveexec_b64 s[{SEXEC}:{SEXEC + 1}], vcc")
        # The lane pair (l, l + 32) holds the two 8-byte halves of 16 contiguous output bytes (d = 8 g + 4 hh + 0..3).
        # One v_permlane32_swap per register pair regroups them so that lane l stores the whole 16 bytes of the even
        # group and lane l + 32 those of the odd group: 2 x global_store_dwordx4 per head-dim block instead of 4 x
        # dwordx2 (the store tail is issue-bound).  Every block converts into registers of its own (qb 0: the S[1]
        # region, qb 1: MINIT, dead by now), so no store has to be waited for before the next block's values are formed.
        obuf = S[1] if qb == 0 else MINIT[0]
        for db in range(4):
            o8 = obuf + 8 * db
            for i in range(16):
                e(f"v_accvgpr_read_b32 {vr(S[0] + i)}, {ar(A_O(qb, db) + i)}")
            e("s_nop 1")
            for i in range(16):
                e(f"v_mul_f32 {vr(S[0] + i)}, {vr(S[0] + i)}, {vr(DLT[0])}")
            for i in range(8):
                e(f"v_cvt_pk_bf16_f32 {vr(o8 + i)}, {vr(S[0] + 2 * i)}, {vr(S[0] + 2 * i + 1)}")
            e("s_nop 1")
            for pr in range(2):       # groups (2 pr, 2 pr + 1): registers o8 + 4 pr + {0, 1 | 2, 3}
                for j in range(2):
                    e(f"v_permlane32_swap_b32 {vr(o8 + 4 * pr + j)}, {vr(o8 + 4 * pr + 2 + j)}")
            e("s_nop 1")
            for pr in range(2):
                if "nostore" in ABL and (db or pr):
                    continue
                e(f"global_store_dwordx4 {vr(OP[qb], 2)}, {vr(o8 + 4 * pr, 4)}, off offset:{db * 64 + pr * 32}")
        e(f"s_mov_b64 exec, s[{SEXEC}:{SEXEC + 1}]")
    end_l = ".Lr64_end%="
    e(f"s_branch {end_l}")
    for ln in cold:
        e(ln)
    e(f"{end_l}:")
    body_txt = "\n".join('    "' + ln.replace("\n\t", '\\n\\t') + '\\n"' for ln in out)
    clob_v = ", ".join(f'"v{i}"' for i in range(1, 256))
    clob_a = ", ".join(f'"a{i}"' for i in range(0, 256))
    clob_s = ", ".join(f'"s{i}"' for i in range(60, 80))
    path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_attention_r64.py -- do not edit; see that file for the design.\n")
        f.write(f"#define SF_R64_N_PARAM {N_PARAM}\n")
        f.write("#define SF_R64_ASM_BODY \\\n" + body_txt.replace("\n", " \\\n") + "\n")
        f.write(f"#define SF_R64_CLOBBERS {clob_v}, {clob_a}, {clob_s}, \"vcc\", \"scc\", \"memory\"\n")
    print(f"wrote {path}: {len(out)} asm lines")


if __name__ == "__main__":
    main()
