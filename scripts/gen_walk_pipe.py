#!/usr/bin/env python3
"""Generates bito_amd/csrc/walk_pipe_gen.inc: the two hand-scheduled gfx950 loops of
walk_pipe_kernel (bito_amd/csrc/walk_pipe.hip) as inline-assembly text.

Why generated assembly.  The traversal is bound by per-step latency and instruction issue, not by
arithmetic: one wave per SIMD (LDS capacity) means nothing hides a dependent
instruction's latency unless the instruction stream itself is software-pipelined across tree steps, and
the compiler cannot do that across the dynamic dispatch on the kinds of a step's children.  Here every
step body is straight-line code with

  * the NEXT step's operands (child messages out of LDS, tip masks, matrix images out of L2, the step
    descriptor) requested while THIS step's arithmetic runs,
  * independent work placed in the shadow of every v_mfma_f64_4x4x4_4b (16 cycles of the matrix pipe),
  * the wait states the hardware needs between a matrix instruction and a dependent read inserted by
    this script (table below, measured from what hipcc itself inserts for gfx950).

The arithmetic per step is documented in walk_pipe.hip.  Register numbers are fixed here; the C++ side
passes a few operands by name and declares the rest clobbered.

usage: python3 scripts/gen_walk_pipe.py   (rewrites bito_amd/csrc/walk_pipe_gen.inc)
"""
import os
import sys

GROUPS = (1, 2, 4)

# ---- wait states (number of instruction issue slots between producer and consumer) -----------------
# measured from hipcc output for gfx950 (v_mfma_f64_4x4x4_4b = "DGEMM 4x4", 4 passes):
WS_MFMA_VALU = 6      # matrix result -> VALU read / VALU overwrite
WS_MFMA_MFMA_AB = 6   # matrix result -> matrix A/B operand
WS_MFMA_MFMA_C = 4    # matrix result -> matrix C operand
WS_MFMA_MEM = 9       # matrix result -> store data / address of a memory or LDS instruction
WS_VALU_MFMA = 2      # VALU result -> matrix operand
SETPC_WAIT_STATES = 4 # what the 21 cycles of a taken s_setpc_b64 count for

VBASE = 32   # first VGPR the loops may use (v0..v31 and a224.. stay with the compiler)
VLIMIT = 256
SBASE, SLIMIT = 32, 100  # SGPRs (s0..s31 stay with the compiler)


import re

_SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b|\b(exec|m0|vcc)\b|%\[(\w+)\]")


def _scalar_operands(text):
    """scalar resources named by an instruction, in operand order: SGPR numbers, 'exec', 'm0', 'vcc', and the
    asm statement's named operands (inputs: never written by the loops)"""
    body = text.split(None, 1)[1] if " " in text else ""
    out = []
    for m in _SREG.finditer(body):
        if m.group(1):
            out.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
        elif m.group(3):
            out.append({int(m.group(3))})
        elif m.group(4):
            out.append({m.group(4)})
        else:
            out.append({"%" + m.group(5)})
    return out


class Ins:
    __slots__ = ("text", "kind", "reads", "writes", "mem_reads", "reads_c", "sreads", "swrites", "fence")

    def __init__(self, text, kind, reads=(), writes=(), mem_reads=(), reads_c=(), indexed=False):
        self.text, self.kind = text, kind
        self.reads, self.writes, self.mem_reads, self.reads_c = list(reads), list(writes), list(mem_reads), list(reads_c)
        self.fence = False
        ops = _scalar_operands(text)
        op = text.split()[0]
        sr, sw = set(), set()
        if kind == "salu":
            no_dest = op.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_nop", "s_set_gpr_idx", "s_cbranch", "s_branch", "s_setpc"))
            for k, o in enumerate(ops):
                (sw if (k == 0 and not no_dest) else sr).update(o)
            if op.startswith("s_set_gpr_idx"):
                sw.update({"m0", "mode"})
            if op == "s_movrels_b32":
                sr.add("m0")
                sr.update(range(0, 104))  # (relative source: any SGPR)
            if op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_setpc")):
                self.fence = True
            if op == "s_nop":  # (only the one between a write of M0 and s_movrels: it stays between them)
                sr.add("m0")
                sw.add("m0")
        else:
            for o in ops:
                sr.update(o)
            if op.startswith("s_load"):  # destination first
                sw.update(ops[0])
                sr.difference_update(ops[0])
            if op.startswith("v_readlane"):
                sw.update(ops[0])
                sr.difference_update(ops[0])
            sr.update({"exec", "mode"})  # every vector / memory instruction runs under EXEC and the index mode
            if op.startswith("global_load_lds"):
                sr.add("m0")  # (the LDS address of a load that goes straight to LDS)
            if indexed:
                sr.add("m0")  # ... and inside an index-mode region M0 is part of its operand
        self.sreads, self.swrites = sr, sw

    def conflicts_with_delayed(self, s):
        """may this instruction NOT be emitted before the delayed scalar instruction s (earlier in program order)?"""
        return bool(self.sreads & s.swrites) or bool(self.swrites & (s.sreads | s.swrites))


class Emitter:
    """Collects instructions; finish() sinks scalar instructions behind matrix instructions (a lone wave
    hides about two SALU instructions behind one v_mfma_f64_4x4x4_4b, nothing else), then inserts the
    s_nop wait states the hazards above need."""

    SALU_PER_MFMA = 2

    def __init__(self):
        self.items = []   # Ins | ("label", name) | ("raw", text) | ("comment", text)
        self.count = {}

    def _raw(self, text):
        self.items.append(("raw", text))

    def comment(self, text):
        self.items.append(("comment", text))

    def label(self, name):
        self.items.append(("label", name))

    def control(self, text, settled=False):
        """branch / jump: nothing moves across it.  settled: the code at the target reads no matrix result and
        feeds no matrix instruction before it has waited, so the hazard bookkeeping just carries on with the
        fall-through path"""
        self.items.append(("control_settled" if settled else "control", text))

    def ins(self, text, kind, reads=(), writes=(), mem_reads=(), reads_c=(), indexed=False):
        self.items.append(Ins(text, kind, reads, writes, mem_reads, reads_c, indexed))
        self.count[kind] = self.count.get(kind, 0) + 1

    # ---- pass 1: scalar instructions into the shadow of matrix instructions ----
    def _schedule(self):
        out, queue = [], []

        def flush(upto=None):
            n = len(queue) if upto is None else upto
            out.extend(queue[:n])
            del queue[:n]

        for it in self.items:
            if not isinstance(it, Ins):
                flush()
                out.append(it)
                continue
            if it.kind == "salu":
                if it.fence or os.environ.get("PIPE_NO_SINK"):
                    flush()
                    out.append(it)
                else:
                    queue.append(it)
                continue
            # vector / memory instruction: everything it depends on leaves the queue first (in order)
            last = -1
            for k, sq in enumerate(queue):
                if it.conflicts_with_delayed(sq):
                    last = k
            if last >= 0:
                flush(last + 1)
            out.append(it)
            if it.kind == "mfma":
                flush(min(self.SALU_PER_MFMA, len(queue)))
        flush()
        return out

    # ---- pass 2: hazards ----
    def finish(self):
        lines = []
        pos = 0
        writer = {}
        last_mfma = last_valu = -100

        def nop(states):
            nonlocal pos
            while states > 0:
                n = min(states, 16)
                lines.append(f"s_nop {n - 1}")
                pos += n
                states -= n

        def require(regs, need_of):
            worst = 0
            for r in regs:
                w = writer.get(r)
                if w is None:
                    continue
                p, kind = w
                need = need_of.get(kind, 0)
                have = pos - p - 1
                worst = max(worst, need - have)
            if worst > 0:
                nop(worst)

        def drain(credit=0):
            need = max(WS_MFMA_MEM - (pos - last_mfma - 1), WS_VALU_MFMA - (pos - last_valu - 1), 0) - credit
            if need > 0:
                nop(need)
            writer.clear()

        for it in self._schedule():
            if not isinstance(it, Ins):
                tag, text = it
                if tag == "comment":
                    lines.append(f"; {text}")
                elif tag == "label":
                    drain()
                    lines.append(f"{text}:")
                elif tag == "control":
                    # (a jump keeps the wave from issuing for 21 cycles, measured: worth four wait states)
                    drain(credit=SETPC_WAIT_STATES if text.startswith("s_setpc") else 0)
                    lines.append(text)
                    pos += 1
                elif tag == "control_settled":
                    lines.append(text)
                    pos += 1
                else:
                    lines.append(text)
                continue
            if it.kind == "mfma":
                require(it.reads, {"mfma": WS_MFMA_MFMA_AB, "valu": WS_VALU_MFMA})
                require(it.reads_c, {"mfma": WS_MFMA_MFMA_C, "valu": WS_VALU_MFMA})
                require(it.writes, {"mfma": WS_MFMA_VALU})
            elif it.kind == "valu":
                require(it.reads, {"mfma": WS_MFMA_VALU})
                require(it.writes, {"mfma": WS_MFMA_VALU})
            elif it.kind == "mem":
                require(it.mem_reads, {"mfma": WS_MFMA_MEM})
                require(it.writes, {"mfma": WS_MFMA_MEM})  # a load landing in a register a matrix op still owns
            lines.append(it.text)
            for r in it.writes:
                if it.kind in ("mfma", "valu"):
                    writer[r] = (pos, it.kind)
                else:
                    writer.pop(r, None)
            if it.kind == "mfma":
                last_mfma = pos
            if it.kind == "valu":
                last_valu = pos
            pos += 1
        self.lines = lines
        return lines


def vp(r):  # 64-bit VGPR pair
    return f"v[{r}:{r + 1}]"


def ap(r):
    return f"a[{r}:{r + 1}]"


class Alloc:
    def __init__(self, base, limit, what):
        self.next, self.limit, self.what = base, limit, what
        self.names = []

    def get(self, n, name, align=1):
        self.next = (self.next + align - 1) // align * align
        r = self.next
        self.next += n
        if self.next > self.limit:
            raise RuntimeError(f"out of {self.what} registers at {name}")
        self.names.append((name, r, n))
        return r


class Loops:
    """One instance per group count G (1, 2 or 4).

    What a lone wave pays per instruction on gfx950 (one wave per SIMD, measured with a microbenchmark of
    1024 repetitions): SALU 4.5 cycles, 32-bit VALU 4.8, v_mul_f64 5.5, v_readlane_b32 9.2, the matrix
    instruction 16.7 back to back -- and a VALU instruction does NOT overlap a matrix instruction of the
    same wave (the pair costs 29 cycles), while SALU instructions disappear behind it (matrix + 2 SALU =
    17.6).  A wait state (s_nop) is 4.5 cycles, a taken branch 21.  ds_read2st64_b64 / ds_write2st64_b64
    move 1 KB per wave at 128 / 79 bytes per clock per CU, ds_read_b128 twice as fast.  Hence:

      * everything that can be scalar is scalar and sits behind matrix instructions: a step's descriptor
        arrives by ONE s_load_dwordx16, already unpacked, into one of two register sets (every body exists once
        per set), requested a step ahead (pre-order) or two (post-order) from a table the unit has warmed into
        the scalar cache; bodies are branch-free and specialised on the kinds of the two children and on
        whether a vector is handed to the next step in registers; the jump to the next body goes through a
        table of code offsets;
      * the images of every branch of the tree live in the AGPR file (P of a tip's branch at a[2 tip], of an
        internal branch behind them, in one of the two layouts below -- loaded once per unit of work) and the matrix instruction
        reads its A operand there through the VGPR index mode (s_set_gpr_idx_on adds M0 to the register number
        of source 0, AGPR sources included): no image is fetched, staged or copied inside the loops;
      * the packed tip masks of all tips of the tile live in VGPRs (one per tip), read through the same
        index mode on source 1 by an SDWA shift -- one vector instruction per tip operand: no tip traffic
        inside the loops either;
      * nothing a body waits for is younger than a step: stores are issued a step late (post-order) or not
        waited for at all, and the edge sums leave for LDS in the middle of matrix instructions;
      * scalar instructions are placed where they are free (pass 1 below sinks them behind matrix instructions;
        the bodies order their own work so that M0, EXEC and index-mode changes fall behind one as well), and the
        wait states the hardware needs are filled with work the body has to do anyway;
      * stored cells are 16 bytes per lane (two pattern groups side by side): ds_read_b128 / ds_write_b128."""

    # Matrix images in the AGPR file, a0..a223 (a224.. and v0..v(VBASE-1) stay with the compiler), in one of two
    # layouts (a step's descriptor names the registers of the images it uses; the post-order loop is the same for
    # both, the pre-order loop exists once per layout):
    #   * up to 38 taxa, EXACT: P of a tip's branch at a[2 tip] (nothing is propagated below a tip, so no P^T),
    #     (P, P^T) of the branch above internal node n + j at a[EXACT_BASE + 4 j]; the pre-order loop multiplies
    #     by the second half where the reference's recursion has P^T (and by the first where it rebuilds a
    #     cherry's message).
    #   * 39 to 48 taxa, ONE image per branch (P at a[2 tip], at a[REV_BASE + 2 j]): the models are reversible,
    #     pi_i P_ij = pi_j P_ji, i.e. P^T = Pi P Pi^-1, so with the pre-order partials kept as v = u / pi the
    #     recursion pre(child) = P^T (u . m) reads v(child) = P (v . m) -- the image of the post-order pass, no
    #     extra multiplication; the root starts from 1 instead of pi and the Q image carries the factor (rows of
    #     r_c Q scaled by pi_i).  The computed P is reversible to rounding only, relative to ITS OWN entries'
    #     rounding error: an entry that is all rounding error -- a zero-length branch -- breaks the identity, and a
    #     pattern that needs a substitution on such a branch then gets derivatives that differ from the
    #     reference's (which are noise there too, but the same noise as the checker's).  The engine therefore uses
    #     this layout only for batches whose branch lengths are all 9e-7 or more (relative error of an
    #     off-diagonal entry <= 1e-10) and keeps the exact one wherever it fits.
    MAX_TIPS = 48      # a VGPR per tip for its packed masks beside one or two pattern groups (64: 5.5 % slower at 36 taxa)
    EXACT_TAXA = 38
    EXACT_BASE = 2 * EXACT_TAXA
    REV_BASE = 2 * MAX_TIPS
    MAX_INNER = MAX_TIPS - 2
    IMAGE_REGS = 224
    assert EXACT_BASE + 4 * (EXACT_TAXA - 2) <= IMAGE_REGS and REV_BASE + 2 * MAX_INNER <= IMAGE_REGS

    #   * 49 to 56 taxa, WIDE (round 3): the one-image layout with room for 56 tips -- 64 mask registers per lane
    #     (5.5 % slower where 48 would do, hence a layout of its own), P of a tip's branch at a[2 tip], of an internal
    #     branch at a[WIDE_REV_BASE + 2 j]: 4 n - 4 = 220 of the 224 image registers at 56 taxa.  One or two pattern
    #     groups per wave only.
    WIDE_MAX_TIPS = 64
    WIDE_TIP_SLOTS = 64
    WIDE_REV_BASE = 2 * WIDE_MAX_TIPS
    WIDE_MAX_INNER = WIDE_MAX_TIPS - 2
    WIDE_IMAGE_REGS = 252  # (the wide kernels carry at most two pattern groups: the compiler needs no AGPR there, checked at build time)
    assert WIDE_REV_BASE + 2 * WIDE_MAX_INNER <= WIDE_IMAGE_REGS

    #   * up to 28 taxa, TWO WAVES PER SIMD (round 4): the one-image layout inside 256 registers per wave -- P of a tip's
    #     branch at a[2 tip], of an internal branch at a[TWO_REV_BASE + 2 j]: 4 n - 4 = 108 AGPRs at 28 taxa, which leaves
    #     148 VGPRs: 28 mask registers, the loops' own 102 beside two pattern groups, and v0..v15 for the compiler.  The
    #     workgroup is eight waves, so a tile's mask rows keep a stride of 32 dwords (a wave copies its rows as whole
    #     dwords per lane) of which a lane loads the first 28.  Why: a lone wave's vector, LDS and scalar instructions
    #     never overlap its own matrix instructions; a sibling wave's do (profiles/r2_issue_costs.txt, p_mix_mix).
    # (PIPE_TWO_TIPS / PIPE_TWO_VBASE: measurement builds -- fewer taxa leave the compiler more of the 256 registers)
    TWO_MAX_TIPS = int(os.environ.get("PIPE_TWO_TIPS", 28))
    TWO_TIP_SLOTS = 32 if TWO_MAX_TIPS > 16 else 16
    TWO_REV_BASE = 2 * TWO_MAX_TIPS
    TWO_MAX_INNER = TWO_MAX_TIPS - 2
    TWO_IMAGE_REGS = TWO_REV_BASE + 2 * TWO_MAX_INNER
    TWO_VBASE = int(os.environ.get("PIPE_TWO_VBASE", 16))
    TWO_VLIMIT = 256 - TWO_IMAGE_REGS
    TWO_WAVES = 8

    #   * 33 to 38 taxa with FOUR pattern groups per wave (round 4, "many"): the exact layout with 40 mask registers beside
    #     the four groups' own -- v32..v253 -- in mask rows of 48 dwords (a wave copies its rows as whole dwords per lane).
    #     Up to 32 taxa keep the 32-register loops (every mask register is an instruction in a loop's prologue).
    MANY_TIP_REGS = 40
    MANY_TIP_SLOTS = 48
    MANY_MAX_TIPS = 38

    def __init__(self, G, exact=True, wide=False, two=False, many=False):
        self.G = G
        self.many = many
        assert not many or (G == 4 and exact and not wide and not two)
        self.exact = exact  # image layout the pre-order loop is generated for (the only place the loops differ)
        self.wide = wide
        self.two = two
        self.waves = 4
        self.vbase, vlimit = VBASE, VLIMIT
        self.image_regs = self.WIDE_IMAGE_REGS if wide else self.IMAGE_REGS
        if wide:  # (instance attributes shadow the class's: every use below goes through self)
            assert G < 4 and not exact
            self.MAX_TIPS, self.REV_BASE, self.MAX_INNER = self.WIDE_MAX_TIPS, self.WIDE_REV_BASE, self.WIDE_MAX_INNER
        if two:
            assert G < 4 and not exact and not wide
            self.MAX_TIPS, self.REV_BASE, self.MAX_INNER = self.TWO_MAX_TIPS, self.TWO_REV_BASE, self.TWO_MAX_INNER
            self.vbase, vlimit, self.image_regs, self.waves = self.TWO_VBASE, self.TWO_VLIMIT, self.TWO_IMAGE_REGS, self.TWO_WAVES
        self.e = None
        V = Alloc(self.vbase, vlimit, "VGPR")
        S = Alloc(SBASE, SLIMIT, "SGPR")
        g2 = lambda name: [V.get(2, f"{name}{g}", 2) for g in range(G)]
        # persistent
        # packed masks of tip t (byte g = mask of this lane's pattern in group g): 32 slots beside four groups'
        # registers, 48 beside fewer
        self.TIP_SLOTS = self.WIDE_TIP_SLOTS if wide else (self.TWO_TIP_SLOTS if two else (self.MANY_TIP_SLOTS if many else (32 if G == 4 else 48)))
        self.TIP_REGS = self.TWO_MAX_TIPS if two else (self.MANY_TIP_REGS if many else self.TIP_SLOTS)  # (the mask registers a lane holds; TIP_SLOTS: its row in LDS)
        self.TMV = V.get(self.TIP_REGS, "TMV", 4)
        self.U = g2("U")                      # pre-order partial of the step's node
        self.ONE = V.get(2, "ONE", 2)
        self.TP = [[V.get(2, f"TP{t}_{g}", 2) for g in range(G)] for t in range(4)]  # tip operands (lo word stays 0)
        self.M = [g2("M0_"), g2("M1_")]       # child messages out of LDS (slot 0, slot 1)
        self.ES = [V.get(2, "ES0", 2), V.get(2, "ES1", 2)]  # per-lane sums of the two child edges (flushed one step late)
        # temporaries
        self.MSG = [g2("MSG0_"), g2("MSG1_")]
        self.MA = [g2("MA0_"), g2("MA1_")]
        self.MB = [g2("MB0_"), g2("MB1_")]
        self.X = [g2("X0_"), g2("X1_")]       # cherry partial ma.mb, later the cherry's pre-order partial
        self.DQ = [g2("DQ0_"), g2("DQ1_")]
        self.W = [g2("W0_"), g2("W1_")]
        # pre-order: the children's partials; post-order: UC[1] = own message.  UC[1] shares X[1]'s registers:
        # X[1] serves a cherry in slot 1, UC[1] a stored cell in slot 1 (post-order: after X[1]'s last use)
        self.UC = [g2("UC0_"), self.X[1]]
        self.R = [V.get(2, "R0", 2), V.get(2, "R1", 2)]
        self.T = [V.get(2, "T0", 2), V.get(2, "T1", 2)]
        # the tail of a cherry slot s (its two tip edges) runs after everything else of the step has been
        # consumed, in registers that are dead by then: DQA = MSG[s], DQB = UC[0] (only a (C,C) step uses
        # it), TA = W[s], TB = DQ[s], EA/EB = T0/T1 (the main pair's stores have been issued)
        self.EA, self.EB = self.T[0], self.T[1]
        self.AD = [V.get(1, f"AD{k}") for k in range(8)]
        self.vnext = V.next
        # scalars
        # two descriptor sets used alternately: a body of parity p reads its step's descriptor in DESC[p],
        # finds the PREVIOUS step's in DESC[1-p] until it requests the NEXT step's into it
        self.DESC = [S.get(16, "DESC0", 4), S.get(16, "DESC1", 4)]
        self.p = 0
        self.TAB = S.get(2, "TAB", 2)
        self.TABOFF = S.get(1, "TABOFF")
        self.WMASK = S.get(2, "WMASK", 2)
        self.BASE = [S.get(2, "BASE0", 2), S.get(2, "BASE1", 2)]  # code address of the bodies of parity 0 / 1
        self.PC = S.get(2, "PC", 2)
        self.OFFTAB = S.get(2 * len(self.KINDS) + 1, "OFFTAB")  # code offsets of the bodies (kinds + their hand-over forms) and of the loop exit
        self.IMGP = S.get(2, "IMGP", 2)   # (the image loader's pointer; the loops use the pair as scratch)
        self.TMP = [self.IMGP, self.IMGP + 1, S.get(1, "TMP2")]
        self.TMPM = [S.get(1, f"TMPM{k}") for k in range(4)]  # scratch of messages() only
        self.snext = S.next
        self.V, self.S = V, S

    # ---- descriptor words (see walk_pipe.hip) ----
    (FLAGS, OWN, NOWN, OFFC0, OFFC1, NOFFC0, NOFFC1, IMG0, IMG1, IMGOWN, TIPA0, TIPB0, TIPA1, TIPB1, E0, E1) = range(16)

    def cur(self, w):
        if self.own_set_requested:
            raise RuntimeError("this step's descriptor set has been handed to the scalar load already")
        return f"s{self.DESC[self.p] + w}"

    def other(self, w):
        return f"s{self.DESC[1 - self.p] + w}"

    # ---- instruction helpers --------------------------------------------------------------------
    idx_mode = None  # None | "SRC0" | "SRC1" while the VGPR index mode is on
    own_set_requested = False  # post-order: the step's own descriptor set already receives the step after next's

    def idx_on(self, sgpr, which):
        self.salu(f"s_set_gpr_idx_on {sgpr}, gpr_idx({which})")
        self.idx_mode = which

    def idx_set(self, sgpr):
        assert self.idx_mode
        self.salu(f"s_set_gpr_idx_idx {sgpr}")

    def idx_off(self):
        self.salu("s_set_gpr_idx_off")
        self.idx_mode = None

    def mfma(self, dst, a, b):
        """a: VGPR pair number, "Q" (the per-tree image operand), or ("A", k): AGPR pair a[k:k+1] offset by
        the index mode (the image of the branch whose index is in M0)"""
        if os.environ.get("PIPE_NO_MFMA"):  # timing experiment: results are wrong
            self.e.ins(f"v_mov_b64 {vp(dst)}, {vp(b)}", "valu", reads=[b, b + 1], writes=[dst, dst + 1])
            return
        if os.environ.get("PIPE_DROP_MFMA"):  # timing experiment (round 4's ablation table): no matrix instruction at all
            return
        if isinstance(a, tuple):
            assert self.idx_mode == "SRC0"
            self.e.ins(f"v_mfma_f64_4x4x4_4b_f64 {vp(dst)}, {ap(a[1])}, {vp(b)}, 0", "mfma", reads=[b, b + 1], writes=[dst, dst + 1],
                       indexed=True)
            return
        assert self.idx_mode is None, "a matrix instruction with a VGPR A operand inside an index-mode region"
        if a == "Q":
            self.e.ins(f"v_mfma_f64_4x4x4_4b_f64 {vp(dst)}, %[q], {vp(b)}, 0", "mfma", reads=[b, b + 1], writes=[dst, dst + 1])
        else:
            self.e.ins(f"v_mfma_f64_4x4x4_4b_f64 {vp(dst)}, {vp(a)}, {vp(b)}, 0", "mfma", reads=[a, a + 1, b, b + 1],
                       writes=[dst, dst + 1])

    def valu(self, text, reads, writes, indexed_ok=False):
        assert self.idx_mode is None or indexed_ok, f"VALU inside an index-mode region: {text}"
        # timing experiments (results are wrong): no FP64 vector arithmetic / no tip operands (the address arithmetic stays)
        if os.environ.get("PIPE_DROP_F64") and text.startswith(("v_mul_f64", "v_fma_f64", "v_mov_b64")):
            return
        if os.environ.get("PIPE_DROP_TIPS") and "sdwa" in text:
            return
        self.e.ins(text, "valu", reads=list(reads), writes=list(writes), indexed=self.idx_mode is not None)

    def vmul(self, dst, a, b):
        self.valu(f"v_mul_f64 {vp(dst)}, {vp(a)}, {vp(b)}", [a, a + 1, b, b + 1], [dst, dst + 1])

    def vfma(self, dst, a, b, c):
        self.valu(f"v_fma_f64 {vp(dst)}, {vp(a)}, {vp(b)}, {vp(c)}", [a, a + 1, b, b + 1, c, c + 1], [dst, dst + 1])

    def vmov64(self, dst, src):
        self.valu(f"v_mov_b64 {vp(dst)}, {vp(src)}", [src, src + 1], [dst, dst + 1])

    def v32(self, text, reads, writes):
        self.valu(text, reads, writes)

    def salu(self, text):
        self.e.ins(text, "salu")

    def mem(self, text, mem_reads=(), writes=()):
        if os.environ.get("PIPE_NO_LDS") and text.startswith("ds_"):  # timing experiment: results are wrong
            return
        self.e.ins(text, "mem", mem_reads=list(mem_reads), writes=list(writes))

    def wait(self, vm=None, lgkm=None):
        parts = []
        if vm is not None:
            parts.append(f"vmcnt({vm})")
        if lgkm is not None:
            parts.append(f"lgkmcnt({lgkm})")
        self.salu("s_waitcnt " + " ".join(parts))

    def label(self, name):
        self.e.label(name)

    def branch(self, cond, target, settled=False):
        self.e.control(f"s_cbranch_{cond} {target}" if cond else f"s_branch {target}", settled)

    def L(self, name):
        return f".Lwp_{self.tag}_{name}_%="

    # EXEC = all lanes when bit `bit` of the next step's flags is set, else none
    def predicate(self, bit):
        self.salu(f"s_bitcmp1_b32 {self.cur(self.FLAGS)}, {bit}")
        self.salu("s_cselect_b64 exec, -1, 0")

    def unpredicate(self):
        self.salu("s_mov_b64 exec, -1")

    # cells: 16 bytes per lane = two pattern groups; pair p at + p * 1024
    def cell_read(self, dst_list, addr_vgpr):
        G = self.G
        n = 0
        for p in range((G + 1) // 2):
            d = dst_list[2 * p]
            if 2 * p + 1 < G:
                if dst_list[2 * p + 1] != d + 2:
                    raise RuntimeError("group registers must be consecutive")
                self.mem(f"ds_read_b128 v[{d}:{d + 3}], v{addr_vgpr} offset:{1024 * p}", mem_reads=[addr_vgpr],
                         writes=list(range(d, d + 4)))
            else:
                self.mem(f"ds_read_b64 {vp(d)}, v{addr_vgpr} offset:{1024 * p}", mem_reads=[addr_vgpr], writes=[d, d + 1])
            n += 1
        return n

    def cell_write(self, src_list, addr_vgpr):
        G = self.G
        n = 0
        for p in range((G + 1) // 2):
            d = src_list[2 * p]
            if 2 * p + 1 < G:
                if src_list[2 * p + 1] != d + 2:
                    raise RuntimeError("group registers must be consecutive")
                self.mem(f"ds_write_b128 v{addr_vgpr}, v[{d}:{d + 3}] offset:{1024 * p}",
                         mem_reads=[addr_vgpr] + list(range(d, d + 4)))
            else:
                self.mem(f"ds_write_b64 v{addr_vgpr}, {vp(d)} offset:{1024 * p}", mem_reads=[addr_vgpr, d, d + 1])
            n += 1
        return n

    def cell_ops(self):  # LDS instructions per cell access
        return (self.G + 1) // 2

    # tip operands of all groups from the packed masks of the tip whose id is in `tip_sgpr`: 2.0 where the
    # state is allowed (the hi word's bit 30), else 0.  A lane's copy of a packed mask keeps only the bit of the
    # lane's own state in every byte (specialise_masks), so byte g shifted by 30 - state is the hi word: one
    # SDWA instruction per group
    def tip_operands(self, requests, leave_on=False):
        """requests: [(tip operand slot, SGPR with the tip id)] -- one index-mode region for all of them;
        leave_on: the caller switches the region to its own mode"""
        for k, (slot_tip, tip_sgpr) in enumerate(requests):
            if k == 0:
                self.idx_on(tip_sgpr, "SRC1")
            else:
                self.idx_set(tip_sgpr)
            for g in range(self.G):
                hi = self.TP[slot_tip][g] + 1
                self.valu(f"v_lshlrev_b32_sdwa v{hi}, %[sh0], v{self.TMV} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD "
                          f"src1_sel:BYTE_{g}", list(range(self.TMV, self.TMV + self.TIP_REGS)), [hi], indexed_ok=True)
        if requests and not leave_on:
            self.idx_off()

    def specialise_masks(self):
        """TMV[t] &= 0x01010101 << state: what is left of byte g is the bit tip_operands shifts to bit 30"""
        k = self.AD[5]
        self.v32(f"v_sub_u32 v{k}, 30, %[sh0]", [], [k])
        self.v32(f"v_mov_b32 v{self.AD[4]}, 0x01010101", [], [self.AD[4]])
        self.v32(f"v_lshlrev_b32 v{k}, v{k}, v{self.AD[4]}", [k, self.AD[4]], [k])
        for t in range(min(self.TIP_REGS, self.MAX_TIPS)):
            self.v32(f"v_and_b32 v{self.TMV + t}, v{k}, v{self.TMV + t}", [k, self.TMV + t], [self.TMV + t])

    # ---- edge sums: 64 lanes -> the gradient row entries of the four blocks (rate categories) ----
    # One matrix instruction sums over the states (R_b[i][j] = sum_k E[16k+4b+i], the same for every j); the lanes
    # with j = 0 -- sixteen, four per block -- then add their values to the block's row entry in LDS (ds_add_f64:
    # four lanes per address, which the LDS unit takes one after the other, in lane order).
    def flush_stage1(self, ea, eb):
        if os.environ.get("PIPE_FLUSH_ALL_LANES"):  # experiment: every lane adds its own value (16 per address)
            self.flush_src = (ea, eb)
            return
        self.flush_src = (self.R[0], self.R[1])
        self.mfma(self.R[0], ea, self.ONE)
        self.mfma(self.R[1], eb, self.ONE)

    def flush_addresses(self, off0, off1):
        """LDS addresses of the two gradient-row entries the next flush adds to (scalar byte offsets)"""
        self.v32(f"v_add_u32 v{self.AD[6]}, {off0}, %[grow]", [], [self.AD[6]])
        self.v32(f"v_add_u32 v{self.AD[7]}, {off1}, %[grow]", [], [self.AD[7]])

    def flush_stage3(self):
        r0, r1 = self.flush_src
        all_lanes = bool(os.environ.get("PIPE_FLUSH_ALL_LANES"))
        if not all_lanes:
            self.salu(f"s_mov_b64 exec, s[{self.WMASK}:{self.WMASK + 1}]")
        # (accumulate: the rows are summed over the tiles of a run)
        self.mem(f"ds_add_f64 v{self.AD[6]}, {vp(r0)}", mem_reads=[self.AD[6], r0, r0 + 1])
        self.mem(f"ds_add_f64 v{self.AD[7]}, {vp(r1)}", mem_reads=[self.AD[7], r1, r1 + 1])
        if not all_lanes:
            self.salu("s_mov_b64 exec, -1")
        return 2

    # ---- descriptor pipeline ----
    def request_descriptor(self):
        """the next step's descriptor into the other set (one scalar load; nothing waits for it before the
        end of the body -- the body finds what it must know about the next step in its own descriptor)"""
        d = self.DESC[1 - self.p]
        # (the offset is advanced BEHIND the load: that addition can hide behind a matrix instruction)
        self.mem(f"s_load_dwordx16 s[{d}:{d + 15}], s[{self.TAB}:{self.TAB + 1}], s{self.TABOFF}")
        self.salu(f"s_add_u32 s{self.TABOFF}, s{self.TABOFF}, 64")

    def request_descriptor_after_next(self):
        """post-order: the descriptor of the step AFTER the next one into this step's own set, once the body has
        read the last of its own fields -- it is waited for half a step later, in the next body, together
        with that body's operands, so that no body waits for a scalar load or a store it has just issued"""
        d = self.DESC[self.p]
        # (the offset is advanced BEHIND the load: that addition can hide behind a matrix instruction)
        self.mem(f"s_load_dwordx16 s[{d}:{d + 15}], s[{self.TAB}:{self.TAB + 1}], s{self.TABOFF}")
        self.salu(f"s_add_u32 s{self.TABOFF}, s{self.TABOFF}, 64")
        self.own_set_requested = True

    def next_pc(self, finish=True, between=None):
        """code address of the next step's body (this step's flags, bits 4..7; the body of the other parity)
        into PC; the jump itself comes last.  finish=False: the table look-up only (it needs M0 and so must
        stand in front of an index-mode region); next_pc_finish() then adds the block's address where the two
        additions can hide behind matrix instructions"""
        t = self.TMP[0]
        self.salu(f"s_bfe_u32 m0, {self.cur(self.FLAGS)}, 0x40004")
        if between is not None:
            between()  # (an instruction the body needs anyway stands in for the wait state behind the M0 write)
        else:
            self.salu("s_nop 0")
        self.salu(f"s_movrels_b32 s{t}, s{self.OFFTAB}")
        if finish:
            self.next_pc_finish()

    def next_pc_finish(self):
        t = self.TMP[0]
        base = self.BASE[1 - self.p]
        self.salu(f"s_add_u32 s{self.PC}, s{base}, s{t}")
        self.salu(f"s_addc_u32 s{self.PC + 1}, s{base + 1}, 0")

    def go_first(self, delayed_store=False):
        """from the loop prologue into the first step (parity 0): its descriptor has landed in DESC[0], its
        stored operands are requested here"""
        ad = self.AD
        t = self.TMP
        self.p = 0
        self.own_set_requested = False
        self.wait(lgkm=0)
        self.specialise_masks()
        self.v32(f"v_add_u32 v{ad[0]}, {self.cur(self.OFFC0)}, %[arena]", [], [ad[0]])
        self.v32(f"v_add_u32 v{ad[1]}, {self.cur(self.OFFC1)}, %[arena]", [], [ad[1]])
        if delayed_store:
            # every post-order body first stores the PREVIOUS step's message; the first one stores whatever is
            # in those registers into its own node's cell, which the real message overwrites a step later
            self.v32(f"v_add_u32 v{ad[6]}, {self.cur(self.OWN)}, %[arena]", [], [ad[6]])
        self.cell_read(self.M[0], ad[0])
        self.cell_read(self.M[1], ad[1])
        self.salu(f"s_and_b32 m0, {self.cur(self.FLAGS)}, 15")
        self.salu("s_nop 0")
        self.salu(f"s_movrels_b32 s{t[0]}, s{self.OFFTAB}")
        self.salu(f"s_add_u32 s{self.PC}, s{self.BASE[0]}, s{t[0]}")
        self.salu(f"s_addc_u32 s{self.PC + 1}, s{self.BASE[0] + 1}, 0")
        self.wait(lgkm=0)
        self.go()

    def go(self):
        self.e.control(f"s_setpc_b64 s[{self.PC}:{self.PC + 1}]")

    def loop_entry(self, names, two_ahead=False):
        """common prologue: first descriptor (post-order: the first two), constants, the tile's tip masks, the
        table of body offsets (relative to the start of a parity's block of bodies; entry 10: that block's
        way out of the loop)"""
        G = self.G
        self.salu(f"s_mov_b64 s[{self.TAB}:{self.TAB + 1}], %[tab]")
        self.salu(f"s_mov_b32 s{self.TABOFF}, {128 if two_ahead else 64}")  # (offset of the next descriptor to request)
        self.mem(f"s_load_dwordx16 s[{self.DESC[0]}:{self.DESC[0] + 15}], s[{self.TAB}:{self.TAB + 1}], 0x0")
        if two_ahead:
            self.mem(f"s_load_dwordx16 s[{self.DESC[1]}:{self.DESC[1] + 15}], s[{self.TAB}:{self.TAB + 1}], 0x40")
        for t in range(4):
            for g in range(G):
                self.v32(f"v_mov_b32 v{self.TP[t][g]}, 0", [], [self.TP[t][g]])
        # packed masks of tip t into TMV[t]: this lane's tip slots are consecutive bytes
        for k in range(self.TIP_REGS // 4):
            self.mem(f"ds_read_b128 v[{self.TMV + 4 * k}:{self.TMV + 4 * k + 3}], %[tiprow] offset:{16 * k}",
                     writes=list(range(self.TMV + 4 * k, self.TMV + 4 * k + 4)))
        self.e.control(f"s_getpc_b64 s[{self.BASE[0]}:{self.BASE[0] + 1}]")
        self.e.label(self.L("here"))
        # BASE[p] = address of parity p's block of bodies
        self.salu(f"s_add_u32 s{self.BASE[0]}, s{self.BASE[0]}, {self.L('block0')}-{self.L('here')}")
        self.salu(f"s_addc_u32 s{self.BASE[0] + 1}, s{self.BASE[0] + 1}, 0")
        self.salu(f"s_add_u32 s{self.BASE[1]}, s{self.BASE[0]}, {self.L('block1')}-{self.L('block0')}")
        self.salu(f"s_addc_u32 s{self.BASE[1] + 1}, s{self.BASE[0] + 1}, 0")
        for k, name in enumerate(names):
            self.salu(f"s_mov_b32 s{self.OFFTAB + k}, {self.L(name + '_0')}-{self.L('block0')}")
        self.salu(f"s_mov_b32 s{self.OFFTAB + 2 * len(self.KINDS)}, {self.L('out_0')}-{self.L('block0')}")

    # ---- child messages of one slot (both passes) ---------------------------------------------------
    def fork(self, kinds):
        """registers and descriptor fields of the pitchfork slot of a step, or None.  The pitchfork's vectors beyond a
        cherry's -- the third tip's message MC and the inner cherry's message MH -- live in the MA / MB registers of the
        OTHER slot, which a tip or a stored cell there leaves idle; its third tip operand in the other slot's spare one."""
        if "F" not in kinds:
            return None
        s = kinds.index("F")
        o = 1 - s
        assert kinds[o] in ("T", "C")
        tip = [[self.TIPA0, self.TIPB0], [self.TIPA1, self.TIPB1]]
        return dict(s=s, MA=self.MA[s], MB=self.MB[s], MC=self.MA[o], MH=self.MB[o], X=self.X[s], MSG=self.MSG[s],
                    TPA=self.TP[2 * s], TPB=self.TP[2 * s + 1], TPC=self.TP[2 * o + 1],
                    tipa=tip[s][0], tipbc=tip[s][1],                    # tip A; the cherry's tips: B | C << 8
                    img=[self.IMG0, self.IMG1][s],                      # image register of the node's branch | of the cherry's << 8
                    edge_h=[self.OFFC0, self.OFFC1][s])                 # gradient-row offset of the cherry's branch (the slot has no cell)

    def messages(self, kinds, scalar_work=None):
        """leaves the message of a tip / cherry / pitchfork slot s in MSG[s] -- a stored cell's message is M[s] itself.
        Cherry: MA, MB (tip messages) are kept for the pre-order pass; pitchfork: MA, MB, MC and the inner cherry's
        message MH.  scalar_work: emits scalar instructions the body needs anyway (they may use M0); they are placed where
        they fill wait states -- between a cherry's tip products and their product -- or, without one, at the end"""
        G = self.G
        t = self.TMPM
        tip = [[self.TIPA0, self.TIPB0], [self.TIPA1, self.TIPB1]]
        img = [self.IMG0, self.IMG1]
        F = self.fork(kinds)
        wanted = []
        if F:  # (the third tip's id out of its packed field first: the index-mode region below reads it)
            self.salu(f"s_lshr_b32 s{t[3]}, {self.cur(F['tipbc'])}, 8")
        for s in (0, 1):
            if kinds[s] == "T":
                wanted.append((2 * s, self.cur(tip[s][0])))
            elif kinds[s] == "H":
                wanted.append((2 * s, self.cur(tip[s][0])))
                wanted.append((2 * s + 1, self.cur(tip[s][1])))
            elif kinds[s] == "F":  # (an index is the low byte of its register: the packed field serves tip B as it is)
                wanted.append((self.TP.index(F["TPA"]), self.cur(F["tipa"])))
                wanted.append((self.TP.index(F["TPB"]), self.cur(F["tipbc"])))
                wanted.append((self.TP.index(F["TPC"]), f"s{t[3]}"))
        self.tip_operands(wanted, leave_on=bool(wanted))
        # one index-mode region for all the tip products (s_set_gpr_idx_on switches the mode from source 1 to
        # source 0 as it sets the index; further indices by s_set_gpr_idx_idx)
        first = True

        def index(sgpr):
            nonlocal first
            if first:
                self.idx_on(sgpr, "SRC0")
            else:
                self.idx_set(sgpr)
            first = False

        for s in (0, 1):
            if kinds[s] == "T":
                index(self.cur(img[s]))
                for g in range(G):
                    self.mfma(self.MSG[s][g], ("A", 0), self.TP[2 * s][g])
            elif kinds[s] == "H":
                # image index of a tip branch = 2 x tip id
                self.salu(f"s_lshl_b32 s{t[0]}, {self.cur(tip[s][0])}, 1")
                self.salu(f"s_lshl_b32 s{t[1]}, {self.cur(tip[s][1])}, 1")
                index(f"s{t[0]}")
                for g in range(G):
                    self.mfma(self.MA[s][g], ("A", 0), self.TP[2 * s][g])
                index(f"s{t[1]}")
                for g in range(G):
                    self.mfma(self.MB[s][g], ("A", 0), self.TP[2 * s + 1][g])
            elif kinds[s] == "F":
                # 2 A, 2 B (the low byte of (B | C << 8) << 1) and 2 C (the low byte of (B | C << 8) >> 7: B < 128)
                self.salu(f"s_lshl_b32 s{t[0]}, {self.cur(F['tipa'])}, 1")
                self.salu(f"s_lshl_b32 s{t[1]}, {self.cur(F['tipbc'])}, 1")
                self.salu(f"s_lshr_b32 s{t[2]}, {self.cur(F['tipbc'])}, 7")
                for sg, dst, tp in ((t[0], F["MA"], F["TPA"]), (t[1], F["MB"], F["TPB"]), (t[2], F["MC"], F["TPC"])):
                    index(f"s{sg}")
                    for g in range(G):
                        self.mfma(dst[g], ("A", 0), tp[g])
        if not first:
            self.idx_off()
        if scalar_work is not None and ("H" in kinds or F):
            scalar_work()
            scalar_work = None
            self.e.comment("(scalar work in the wait states between the tip products and their product)")  # (a comment pins them here)
        if F:  # the inner cherry's image register (after the scalar work: that may use the temporaries' neighbours, not these)
            self.salu(f"s_lshr_b32 s{t[0]}, {self.cur(F['img'])}, 8")
        for s in (0, 1):
            if kinds[s] == "H":
                for g in range(G):
                    self.vmul(self.X[s][g], self.MA[s][g], self.MB[s][g])
            elif kinds[s] == "F":
                for g in range(G):
                    self.vmul(F["X"][g], F["MB"][g], F["MC"][g])  # the inner cherry's partial
        first = True
        for s in (0, 1):
            if kinds[s] == "H":
                index(self.cur(img[s]))
                for g in range(G):
                    self.mfma(self.MSG[s][g], ("A", 0), self.X[s][g])
            elif kinds[s] == "F":
                index(f"s{t[0]}")
                for g in range(G):
                    self.mfma(F["MH"][g], ("A", 0), F["X"][g])  # ... and its message
        if not first:
            self.idx_off()
        if F:  # the pitchfork's own partial and message
            for g in range(G):
                self.vmul(F["X"][g], F["MA"][g], F["MH"][g])
            self.idx_on(self.cur(F["img"]), "SRC0")
            for g in range(G):
                self.mfma(F["MSG"][g], ("A", 0), F["X"][g])
            self.idx_off()
        if scalar_work is not None:
            scalar_work()

    def msg(self, s, kind):
        return self.M[s] if kind == "C" else self.MSG[s]

    # =============================== post-order loop ===============================================
    # body index = position here; the second five hand a vector to the NEXT step in registers (post-order:
    # this node's message is that step's slot-1 operand; pre-order: slot 1's partial is that step's U)
    # Child kinds: T tip, C stored cell, H cherry (two tips; rebuilt, never stored), F PITCHFORK (round 4): a tip and a
    # cherry under one node, folded into its parent's step like a cherry when the node's sibling is a tip or a stored cell
    # (kinds tf, fc: one pitchfork per step, so its extra vectors live in registers the sibling's kind leaves idle and the
    # loops need no register more).  A sixth of a random tree's internal nodes are pitchforks: a DS1 tree keeps 14 vectors
    # instead of 17.5, a 64-taxon tree 33 instead of 41 -- which is what lets it run with two pattern groups per wave.
    KINDS = [("cc", "C", "C"), ("tc", "T", "C"), ("hc", "H", "C"), ("th", "T", "H"), ("hh", "H", "H"), ("tf", "T", "F"),
             ("fc", "F", "C")]
    POST_VARIANTS = [(n, a, b, False) for n, a, b in KINDS] + [(n + "f", a, b, True) for n, a, b in KINDS]
    PRE_VARIANTS = [(n, a, b, False) for n, a, b in KINDS] + [(n + "f", a, b, True) for n, a, b in KINDS if b == "C"]

    def post_body(self, name, K0, K1, hand_over, parity):
        """What a body waits for is a step old: its stored operands were requested by the previous body under
        that body's matrix instructions, its descriptor by the body before that, and its own message is stored
        by the NEXT body (LDS executes a wave's instructions in order, so a parent's read, which is issued
        later still, finds it).  The one s_waitcnt of a body therefore never sees a request younger than the
        tip work in front of it."""
        G = self.G
        self.p = parity
        self.own_set_requested = False
        kinds = (K0, K1)
        held = self.DQ[0]  # the step's message until the next body stores it (registers the pre-order loop owns)
        own = self.M[1] if hand_over else held
        ad = self.AD
        self.label(self.L(f"{name}_{parity}"))
        self.e.comment(f"post-order step, children ({K0},{K1})" + (", message handed to the next step" if hand_over else "")
                       + f", descriptor set {parity}")
        # (the table look-up behind the tip products; without a cherry the step's one wait sits in its middle)
        waited = [False]

        def wait_here():
            self.wait(lgkm=0)
            waited[0] = True

        self.messages(kinds, lambda: self.next_pc(finish=False, between=None if "H" in kinds else wait_here))
        if not waited[0]:
            self.wait(lgkm=0)  # stored operands, the next step's descriptor (and stores two steps old)
        self.cell_write(held, ad[6])  # the previous step's message
        # x = m0 . m1 (the node's partial); the root's leaves the loop in X[0]
        m0, m1 = self.msg(0, K0), self.msg(1, K1)
        for g in range(G):
            self.vmul(self.X[0][g], m0[g], m1[g])
        self.v32(f"v_add_u32 v{ad[0]}, {self.cur(self.NOFFC0)}, %[arena]", [], [ad[0]])
        if not hand_over:
            self.v32(f"v_add_u32 v{ad[1]}, {self.cur(self.NOFFC1)}, %[arena]", [], [ad[1]])
        self.v32(f"v_add_u32 v{ad[6]}, {self.cur(self.OWN)}, %[arena]", [], [ad[6]])
        # (No test for the last step: the root's body runs to its end like any other -- a message nobody reads, from
        # image 0 and cell 0 -- and its jump, body index 10, leaves the loop; a compare and an untaken branch per
        # step cost more than that, 16 cycles each.)
        # own message a = P_v x (straight into the next step's slot-1 registers when it is handed over), the
        # next step's stored operands and the descriptor after it requested underneath
        self.idx_on(self.cur(self.IMGOWN), "SRC0")
        self.request_descriptor_after_next()
        self.next_pc_finish()  # (behind the matrix instructions that follow)
        for g in range(G):
            self.mfma(own[g], ("A", 0), self.X[0][g])
        self.idx_off()
        self.cell_read(self.M[0], ad[0])
        if not hand_over:
            self.cell_read(self.M[1], ad[1])
        if hand_over:
            for g in range(G):
                self.vmov64(held[g], own[g])
        self.go()

    def post_loop(self):
        self.e = Emitter()
        self.tag = f"post{self.G}"
        G = self.G
        e = self.e
        e.comment(f"post-order loop, G = {G}")
        names = [v[0] for v in self.POST_VARIANTS]
        if os.environ.get("PIPE_EMPTY_LOOPS"):  # timing experiment: what a tile costs OUTSIDE the two loops
            self.branch(None, self.L("skip"))
        self.loop_entry(names, two_ahead=True)
        self.go_first(delayed_store=True)
        for parity in (0, 1):
            self.label(self.L(f"block{parity}"))
            for name, K0, K1, hand_over in self.POST_VARIANTS:
                self.post_body(name, K0, K1, hand_over, parity)
            self.label(self.L(f"out_{parity}"))  # (where the root's step jumps to)
            self.branch(None, self.L("root"))
        self.label(self.L("root"))
        self.wait(vm=0, lgkm=0)
        if os.environ.get("PIPE_EMPTY_LOOPS"):
            self.label(self.L("skip"))
        for g in range(G):
            self.e.ins(f"v_mov_b64 %[r{g}], {vp(self.X[0][g])}", "valu", reads=[self.X[0][g], self.X[0][g] + 1])
        return e

    # =============================== pre-order loop ================================================
    def pre_body(self, name, K0, K1, hand_over, parity):
        G = self.G
        self.p = parity
        kinds = (K0, K1)
        t = self.TMP
        ad = self.AD
        img = [self.IMG0, self.IMG1]
        tip = [[self.TIPA0, self.TIPB0], [self.TIPA1, self.TIPB1]]
        self.label(self.L(f"{name}_{parity}"))
        self.e.comment(f"pre-order step, children ({K0},{K1})" + (", slot 1's partial handed to the next step" if hand_over else ""))
        # where the previous step's two edge sums go (its descriptor is still in the other set), then their
        # reduction, level 1; the next body's code address sinks behind the matrix instructions
        self.flush_addresses(self.other(self.E0), self.other(self.E1))
        self.flush_stage1(self.ES[0], self.ES[1])
        # Tip and cherry messages first, then Q m of the stored children (their messages are in registers
        # already), then Q m of the others: a matrix result is not a matrix operand for six wait states, and this
        # order puts other matrix instructions in between.  The previous step's edge sums leave for LDS in the
        # middle of the matrix instructions (behind the first group that follows their own reduction): there the
        # two EXEC moves and the two LDS instructions cost nothing, and the wait below -- for the next descriptor,
        # which must be lgkmcnt(0) -- finds them long done
        self.request_descriptor()
        self.messages(kinds, self.next_pc)  # (the jump address: M0 is free there, and matrix instructions follow)
        flushed = False
        for s in (0, 1):
            if kinds[s] == "C":
                for g in range(G):
                    self.mfma(self.DQ[s][g], "Q", self.M[s][g])
                if not flushed:
                    self.flush_stage3()
                    flushed = True
        if not flushed:
            self.flush_stage3()
        for s in (0, 1):
            if kinds[s] != "C":
                for g in range(G):
                    self.mfma(self.DQ[s][g], "Q", self.MSG[s][g])
        self.wait(lgkm=0)  # U when it came out of LDS; the next step's descriptor
        # w_s = U . (message of the other child)
        m0, m1 = self.msg(0, K0), self.msg(1, K1)
        for g in range(G):
            self.vmul(self.W[0][g], self.U[g], m1[g])
            self.vmul(self.W[1][g], self.U[g], m0[g])
        # Cell addresses: the cells this step's children's partials go to are the cells their messages came from,
        # i.e. the addresses the previous body computed for its requests -- two register pairs used alternately
        # (bodies come once per parity), so a step computes the next step's addresses only
        mine = [ad[0], ad[1]] if parity == 0 else [ad[3], ad[4]]
        nxt = [ad[3], ad[4]] if parity == 0 else [ad[0], ad[1]]
        self.v32(f"v_add_u32 v{nxt[0]}, {self.cur(self.NOFFC0)}, %[arena]", [], [nxt[0]])
        self.v32(f"v_add_u32 v{nxt[1]}, {self.cur(self.NOFFC1)}, %[arena]", [], [nxt[1]])
        if not hand_over:
            self.v32(f"v_add_u32 v{ad[2]}, {self.cur(self.NOWN)}, %[arena]", [], [ad[2]])
        # P w (= P^T of the reference's recursion, in the u / pi variables): the children's pre-order partials, with
        # the next step's reads issued underneath
        # (the registers they land in -- M[0], M[1], U -- have had their last use)
        first = True
        regions = sum(1 for k in kinds if k != "T")

        def next_reads():
            self.cell_read(self.M[0], nxt[0])
            self.cell_read(self.M[1], nxt[1])
            if not hand_over:
                self.cell_read(self.U, ad[2])

        for s in (0, 1):
            if kinds[s] == "T":
                continue
            dst = self.UC[s] if kinds[s] == "C" else self.X[s]
            if s == 1 and hand_over:
                dst = self.U  # (every use of this step's U has been issued)
            if first:
                self.idx_on(self.cur(img[s]), "SRC0")
            else:
                self.idx_set(self.cur(img[s]))
            for g in range(G):
                self.mfma(dst[g], ("A", 2 if self.exact else 0), self.W[s][g])
            if first and regions > 1:
                next_reads()
            first = False
        self.idx_off()  # (right behind the last matrix instruction)
        if regions <= 1:
            next_reads()
        # cherry children: the two tip edges under each (gradient rows 8 x tip id).  The tip messages are spent by
        # then, so the products u . m_other of the two tips overwrite them in place (TA in MB, TB in MA).  A
        # cherry's sums leave for LDS behind the next block of work: a matrix result is not storable for nine
        # wait states -- the second cherry's matrix instructions, or the step's own edge sums, which therefore
        # come last
        nst = 0
        pending = False
        for s in (0, 1):
            if kinds[s] != "H":
                continue
            DQA, DQB, TA, TB = self.MSG[s], self.UC[0], self.MB[s], self.MA[s]
            for g in range(G):
                self.mfma(DQA[g], "Q", self.MA[s][g])
            self.salu(f"s_lshl_b32 s{t[1]}, {self.cur(tip[s][0])}, 3")  # gradient rows of the two tip edges: 8 x tip id
            self.salu(f"s_lshl_b32 s{t[2]}, {self.cur(tip[s][1])}, 3")
            for g in range(G):
                self.mfma(DQB[g], "Q", self.MB[s][g])
            if pending:
                nst += self.flush_stage3()
                pending = False
            for g in range(G):
                self.vmul(TA[g], self.X[s][g], self.MB[s][g])
                self.vmul(TB[g], self.X[s][g], self.MA[s][g])
            for g in range(G):
                if g == 0:
                    self.vmul(self.EA, TA[g], DQA[g])
                    self.vmul(self.EB, TB[g], DQB[g])
                else:
                    self.vfma(self.EA, TA[g], DQA[g], self.EA)
                    self.vfma(self.EB, TB[g], DQB[g], self.EB)
            self.flush_addresses(f"s{t[1]}", f"s{t[2]}")
            self.flush_stage1(self.EA, self.EB)
            pending = True
        # a pitchfork child: the edges of its tip A and of its cherry H, then the cherry's partial and its two tip edges
        # (same forms as a cherry's tail, two levels deep; every product lands in registers that are dead by then:
        # Q m in MSG[s] and UC[0] -- neither slot's stored partial goes there in these kinds --, u . m in place)
        F = self.fork(kinds)
        if F:
            s = F["s"]
            q, DQA, DQH = F["X"], self.MSG[s], self.UC[0]
            tm = self.TMPM
            for g in range(G):
                self.mfma(DQA[g], "Q", F["MA"][g])
            self.salu(f"s_lshl_b32 s{t[1]}, {self.cur(F['tipa'])}, 3")  # gradient row of tip A's edge: 8 x tip id
            for g in range(G):
                self.mfma(DQH[g], "Q", F["MH"][g])
            if pending:
                nst += self.flush_stage3()
                pending = False
            for g in range(G):
                self.vmul(F["MH"][g], q[g], F["MH"][g])  # u . m_H: tip A's edge
                self.vmul(F["MA"][g], q[g], F["MA"][g])  # u . m_A: the cherry's edge, and what its partial is made of
            for g in range(G):
                if g == 0:
                    self.vmul(self.EA, F["MH"][g], DQA[g])
                    self.vmul(self.EB, F["MA"][g], DQH[g])
                else:
                    self.vfma(self.EA, F["MH"][g], DQA[g], self.EA)
                    self.vfma(self.EB, F["MA"][g], DQH[g], self.EB)
            self.flush_addresses(f"s{t[1]}", self.cur(F["edge_h"]))
            self.flush_stage1(self.EA, self.EB)
            self.salu(f"s_lshr_b32 s{tm[0]}, {self.cur(F['img'])}, 8")  # the cherry's image register
            self.idx_on(f"s{tm[0]}", "SRC0")
            for g in range(G):
                self.mfma(q[g], ("A", 2 if self.exact else 0), F["MA"][g])  # the cherry's partial (over the pitchfork's)
            self.idx_off()
            for g in range(G):
                self.mfma(DQA[g], "Q", F["MB"][g])
            self.salu(f"s_and_b32 s{t[1]}, {self.cur(F['tipbc'])}, 0xff")  # rows of the cherry's tips B and C
            self.salu(f"s_lshl_b32 s{t[1]}, s{t[1]}, 3")
            self.salu(f"s_lshr_b32 s{t[2]}, {self.cur(F['tipbc'])}, 8")
            self.salu(f"s_lshl_b32 s{t[2]}, s{t[2]}, 3")
            for g in range(G):
                self.mfma(DQH[g], "Q", F["MC"][g])
            nst += self.flush_stage3()  # (A's and H's sums leave)
            for g in range(G):
                self.vmul(F["MC"][g], q[g], F["MC"][g])  # u_H . m_C: tip B's edge
                self.vmul(F["MB"][g], q[g], F["MB"][g])  # u_H . m_B: tip C's
            for g in range(G):
                if g == 0:
                    self.vmul(self.EA, F["MC"][g], DQA[g])
                    self.vmul(self.EB, F["MB"][g], DQH[g])
                else:
                    self.vfma(self.EA, F["MC"][g], DQA[g], self.EA)
                    self.vfma(self.EB, F["MB"][g], DQH[g], self.EB)
            self.flush_addresses(f"s{t[1]}", f"s{t[2]}")
            self.flush_stage1(self.EA, self.EB)
            pending = True
        # edge sums of this step's two child edges (flushed by the next body)
        for s in (0, 1):
            for g in range(G):
                if g == 0:
                    self.vmul(self.ES[s], self.W[s][g], self.DQ[s][g])
                else:
                    self.vfma(self.ES[s], self.W[s][g], self.DQ[s][g], self.ES[s])
        # this step's stores
        if K0 == "C":
            nst += self.cell_write(self.UC[0], mine[0])
        if K1 == "C":
            nst += self.cell_write(self.U if hand_over else self.UC[1], mine[1])
        if pending:
            nst += self.flush_stage3()
        self.wait(lgkm=nst)  # everything requested has landed; the stores may still travel
        self.go()

    def pre_loop(self):
        self.e = Emitter()
        self.tag = f"pre{self.G}"
        G = self.G
        e = self.e
        e.comment(f"pre-order loop, G = {G}")
        # (body index = kind, + the number of kinds with slot 1's partial handed over: only kinds with a stored cell there)
        names = [n for n, _, _ in self.KINDS] + [n + "f" if b == "C" else "cc" for n, a, b in self.KINDS]
        if os.environ.get("PIPE_EMPTY_LOOPS"):
            self.branch(None, self.L("skip"))
        self.loop_entry(names)
        self.e.ins(f"v_mov_b64 {vp(self.ONE)}, 1.0", "valu", writes=[self.ONE, self.ONE + 1])
        for s in (0, 1):
            self.e.ins(f"v_mov_b64 {vp(self.ES[s])}, 0", "valu", writes=[self.ES[s], self.ES[s] + 1])
        for g in range(G):
            self.e.ins(f"v_mov_b64 {vp(self.U[g])}, %[u{g}]", "valu", writes=[self.U[g], self.U[g] + 1])
        # the first body flushes "the previous step's" edge sums: zeros, into the root's slot
        self.salu(f"s_mov_b32 s{self.DESC[1] + self.E0}, %[rootedge]")
        self.salu(f"s_mov_b32 s{self.DESC[1] + self.E1}, %[rootedge]")
        self.salu(f"s_mov_b32 s{self.WMASK}, 0x11111111")  # lanes 16 i + 4 b: column 0 of every block's row sums
        self.salu(f"s_mov_b32 s{self.WMASK + 1}, 0x11111111")
        self.go_first()
        for parity in (0, 1):
            self.label(self.L(f"block{parity}"))
            for name, K0, K1, hand_over in self.PRE_VARIANTS:
                self.pre_body(name, K0, K1, hand_over, parity)
            # the way out: behind the last step (its descriptor is in the other set) the last two edge sums
            self.label(self.L(f"out_{parity}"))
            self.p = parity
            self.flush_addresses(self.other(self.E0), self.other(self.E1))
            self.branch(None, self.L("done"))
        self.label(self.L("done"))
        self.flush_stage1(self.ES[0], self.ES[1])
        self.flush_stage3()
        self.wait(vm=0, lgkm=0)
        if os.environ.get("PIPE_EMPTY_LOOPS"):
            self.label(self.L("skip"))
        return e

    # =============================== image loader ==================================================
    def load_images(self, exact):
        """the tree's matrix images into the AGPR file of every wave: the four waves of the workgroup fetch a
        quarter of the branches each, straight into LDS (global_load_lds_dwordx4; the arena is idle between two
        units), and after a barrier every wave reads all of them from there -- P of %[ntips] tip branches (8 of
        a lane's 16 bytes) into a[2 tip], and of %[ninner] internal branches (P, P^T) into a[EXACT_BASE + 4 j] (exact
        layout) or P into a[REV_BASE + 2 j].
        Four waves fetching the same 40 to 75 KB through one L1 was 7 k cycles per unit."""
        self.e = Emitter()
        self.tag = "load"
        e = self.e
        e.comment("matrix images of the whole tree into the AGPR file")
        ip = self.IMGP
        scratch = self.DESC[1]  # (the loops' descriptor registers are idle here)
        s_branch, s_m0, s_nb = scratch, scratch + 1, scratch + 2
        # the step tables into the scalar cache: %[lines] descriptors of 64 bytes behind %[tab]
        warm = self.L("warm")
        for k in range(2 * (self.MAX_TIPS + 1)):
            if k % 4 == 0:
                self.salu(f"s_cmp_le_u32 %[lines], {k}")
                self.e.control(f"s_cbranch_scc1 {warm}")
            self.mem(f"s_load_dword s{self.TMP[2]}, %[tab], {hex(64 * k)}")
        self.e.label(warm)
        # global -> LDS: this wave's branches are wave, wave + W, ... (W waves per workgroup)
        W = self.waves
        self.salu(f"s_add_u32 s{s_nb}, %[ntips], %[ninner]")
        self.salu(f"s_lshl_b32 s{s_m0}, %[wave], 10")
        self.salu(f"s_mov_b64 s[{ip}:{ip + 1}], %[img]")
        self.salu(f"s_add_u32 s{ip}, s{ip}, s{s_m0}")
        self.salu(f"s_addc_u32 s{ip + 1}, s{ip + 1}, 0")
        self.salu(f"s_add_u32 s{s_m0}, s{s_m0}, %[stage_s]")
        self.salu(f"s_mov_b32 s{s_branch}, %[wave]")
        staged = self.L("staged")
        for k in range((self.MAX_TIPS + self.MAX_INNER + W - 1) // W):
            self.salu(f"s_cmp_ge_u32 s{s_branch}, s{s_nb}")
            self.e.control(f"s_cbranch_scc1 {staged}")
            self.salu(f"s_mov_b32 m0, s{s_m0}")
            self.salu("s_nop 0")
            self.mem(f"global_load_lds_dwordx4 %[lane16], s[{ip}:{ip + 1}]")
            self.salu(f"s_add_u32 s{s_branch}, s{s_branch}, {W}")
            self.salu(f"s_add_u32 s{s_m0}, s{s_m0}, {1024 * W}")
            self.salu(f"s_add_u32 s{ip}, s{ip}, {1024 * W}")
            self.salu(f"s_addc_u32 s{ip + 1}, s{ip + 1}, 0")
        self.e.label(staged)
        self.wait(vm=0)
        self.e.control("s_barrier")
        # LDS -> AGPRs, every wave all branches
        tips_done = self.L("tips")
        for t in range(self.MAX_TIPS):
            if t % 4 == 0:
                self.salu(f"s_cmp_le_u32 %[ntips], {t}")
                self.e.control(f"s_cbranch_scc1 {tips_done}")
            self.mem(f"ds_read_b64 a[{2 * t}:{2 * t + 1}], %[stage_tips] offset:{t * 1024}")
        self.e.label(tips_done)
        done = self.L("done")
        for j in range(self.EXACT_TAXA - 2 if exact else self.MAX_INNER):
            if j % 4 == 0:
                self.salu(f"s_cmp_le_u32 %[ninner], {j}")
                self.e.control(f"s_cbranch_scc1 {done}")
            if exact:
                r = self.EXACT_BASE + 4 * j
                self.mem(f"ds_read_b128 a[{r}:{r + 3}], %[stage_inner] offset:{j * 1024}")
            else:
                r = self.REV_BASE + 2 * j
                self.mem(f"ds_read_b64 a[{r}:{r + 1}], %[stage_inner] offset:{j * 1024}")
        self.e.label(done)
        self.wait(vm=0, lgkm=0)
        self.e.control("s_barrier")  # (the staging area is the arena: nobody writes a cell before everybody has read)
        return e


def fix_q(lines):
    return [ln.replace("v[Q:Q1]", "%[q]") for ln in lines]


def as_macro(name, lines):
    out = [f"#define {name} \\"]
    for ln in lines:
        text = ln.replace("\\", "\\\\").replace('"', '\\"')
        out.append(f'  "{text}\\n" \\')
    out.append('  ""')
    return "\n".join(out)


def clobbers(loops):
    regs = [f"v{r}" for r in range(loops.vbase, loops.vnext)] + \
           [f"a{r}" for r in range(loops.image_regs)] + \
           [f"s{r}" for r in range(SBASE, loops.snext)]
    return ", ".join(f'"{r}"' for r in regs) + ', "vcc", "scc", "memory"  /* (not m0: clang rejects it on a clobber list as a reserved register; it keeps no value there across a statement) */'


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = ["// GENERATED by scripts/gen_walk_pipe.py -- do not edit; see that script for what and why.",
           "// clang-format off"]
    listing = []
    for G in GROUPS:
        loops = Loops(G)
        post = loops.post_loop()
        pre = loops.pre_loop()
        out.append(as_macro(f"WALK_PIPE_POST_ASM_G{G}", post.finish()))
        out.append(as_macro(f"WALK_PIPE_PRE_ASM_G{G}", pre.finish()))
        if G < 4:  # (39 taxa and more never run with four pattern groups per wave: 32 tip-mask registers there)
            out.append(as_macro(f"WALK_PIPE_PRE_REV_ASM_G{G}", Loops(G, exact=False).pre_loop().finish()))
        out.append(f"#define WALK_PIPE_CLOBBERS_G{G} {clobbers(loops)}")
        out.append(f"#define WALK_PIPE_TIP_SLOTS_G{G} {loops.TIP_SLOTS}")
        listing.append(f"G={G}: VGPR v{loops.vbase}..v{loops.vnext - 1}, AGPR a0..a{Loops.IMAGE_REGS - 1}, SGPR s{SBASE}..s{loops.snext - 1}; "
                       f"post {len(post.lines)} lines {post.count}, pre {len(pre.lines)} lines {pre.count}")
    for G in (1, 2):  # the wide layout: 49 to 56 taxa
        loops = Loops(G, exact=False, wide=True)
        post = loops.post_loop()
        pre = Loops(G, exact=False, wide=True).pre_loop()
        out.append(as_macro(f"WALK_PIPE_POST_W_ASM_G{G}", post.finish()))
        out.append(as_macro(f"WALK_PIPE_PRE_REV_W_ASM_G{G}", pre.finish()))
        out.append(f"#define WALK_PIPE_CLOBBERS_W_G{G} {clobbers(loops)}")
        listing.append(f"G={G} wide: VGPR v{loops.vbase}..v{loops.vnext - 1}; post {len(post.lines)} lines {post.count}, pre {len(pre.lines)} lines {pre.count}")
    out.append(as_macro("WALK_PIPE_LOAD_REV_W_ASM", Loops(1, exact=False, wide=True).load_images(False).finish()))
    out.append(f"#define WALK_PIPE_W_TIP_SLOTS {Loops.WIDE_TIP_SLOTS}")
    out.append(f"#define WALK_PIPE_W_MAX_TIPS {Loops.WIDE_MAX_TIPS}")
    out.append(f"#define WALK_PIPE_W_MAX_INNER {Loops.WIDE_MAX_INNER}")
    out.append(f"#define WALK_PIPE_W_REV_BASE {Loops.WIDE_REV_BASE}")
    out.append(f"#define WALK_PIPE_W_IMAGE_REGS {Loops.WIDE_IMAGE_REGS}")
    # four pattern groups per wave with 40 mask registers: 33 to 38 taxa
    loops = Loops(4, many=True)
    post = loops.post_loop()
    pre = Loops(4, many=True).pre_loop()
    out.append(as_macro("WALK_PIPE_POST_M_ASM_G4", post.finish()))
    out.append(as_macro("WALK_PIPE_PRE_M_ASM_G4", pre.finish()))
    out.append(f"#define WALK_PIPE_CLOBBERS_M_G4 {clobbers(loops)}")
    out.append(f"#define WALK_PIPE_M_TIP_SLOTS {Loops.MANY_TIP_SLOTS}")
    out.append(f"#define WALK_PIPE_M_MAX_TIPS {Loops.MANY_MAX_TIPS}")
    listing.append(f"G=4, 40 mask registers: VGPR v{loops.vbase}..v{loops.vnext - 1}; post {len(post.lines)} lines {post.count}, pre {len(pre.lines)} lines {pre.count}")
    for G in (1, 2):  # two waves per SIMD: up to 28 taxa inside 256 registers per wave
        loops = Loops(G, exact=False, two=True)
        post = loops.post_loop()
        pre = Loops(G, exact=False, two=True).pre_loop()
        out.append(as_macro(f"WALK_PIPE_POST_T_ASM_G{G}", post.finish()))
        out.append(as_macro(f"WALK_PIPE_PRE_REV_T_ASM_G{G}", pre.finish()))
        out.append(f"#define WALK_PIPE_CLOBBERS_T_G{G} {clobbers(loops)}")
        listing.append(f"G={G} two waves per SIMD: VGPR v{loops.vbase}..v{loops.vnext - 1}, AGPR a0..a{loops.image_regs - 1}; "
                       f"post {len(post.lines)} lines {post.count}, pre {len(pre.lines)} lines {pre.count}")
    out.append(as_macro("WALK_PIPE_LOAD_REV_T_ASM", Loops(1, exact=False, two=True).load_images(False).finish()))
    out.append(f"#define WALK_PIPE_T_TIP_SLOTS {Loops.TWO_TIP_SLOTS}")
    out.append(f"#define WALK_PIPE_T_MAX_TIPS {Loops.TWO_MAX_TIPS}")
    out.append(f"#define WALK_PIPE_T_MAX_INNER {Loops.TWO_MAX_INNER}")
    out.append(f"#define WALK_PIPE_T_REV_BASE {Loops.TWO_REV_BASE}")
    out.append(f"#define WALK_PIPE_T_IMAGE_REGS {Loops.TWO_IMAGE_REGS}")
    out.append(f"#define WALK_PIPE_T_VBASE {Loops.TWO_VBASE}")
    loops = Loops(1)
    out.append(as_macro("WALK_PIPE_LOAD_EXACT_ASM", loops.load_images(True).finish()))
    loops = Loops(1)
    out.append(as_macro("WALK_PIPE_LOAD_REV_ASM", loops.load_images(False).finish()))
    out.append(f"#define WALK_PIPE_KINDS {len(Loops.KINDS)}  /* body index = kind (cc tc hc th hh tf fc), + this with a vector handed over; exit = twice this */")
    out.append(f"#define WALK_PIPE_MAX_TIPS {Loops.MAX_TIPS}")
    out.append(f"#define WALK_PIPE_MAX_INNER {Loops.MAX_INNER}")
    out.append(f"#define WALK_PIPE_EXACT_TAXA {Loops.EXACT_TAXA}")
    out.append(f"#define WALK_PIPE_EXACT_BASE {Loops.EXACT_BASE}")
    out.append(f"#define WALK_PIPE_REV_BASE {Loops.REV_BASE}")
    out.append(f"#define WALK_PIPE_IMAGE_REGS {Loops.IMAGE_REGS}")
    out.append("// " + "\n// ".join(listing))
    out.append("// clang-format on")
    path = os.path.join(root, "bito_amd", "csrc", "walk_pipe_gen.inc")
    args = sys.argv[1:]
    if "--out" in args:  # (the currency test writes beside the tracked file, not over it: --out <path>)
        k = args.index("--out")
        path = args[k + 1]
        del args[k:k + 2]
    text = "\n".join(out) + "\n"
    if not (os.path.exists(path) and open(path).read() == text):  # (an unchanged file keeps its time stamp: no rebuild)
        with open(path, "w") as fh:
            fh.write(text)
    print("\n".join(listing))
    if args:  # plain listing of one loop for reading: gen_walk_pipe.py pre4 > /tmp/pre4.s
        want = args[0]
        loops = Loops(int(want[-1]))
        e = loops.post_loop() if want.startswith("post") else (loops.pre_loop() if want.startswith("pre") else loops.load_images(True))
        sys.stderr.write("\n".join(e.finish()) + "\n")


if __name__ == "__main__":
    main()
