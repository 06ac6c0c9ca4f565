// gfx950_asm.hpp -- an interpreter for the gfx950 assembly that scripts/gen_walk_pipe.py generates (the two
// hand-scheduled loops and the image loader of walk_pipe_kernel, bito_amd/csrc/walk_pipe_gen.inc), so that the headline
// kernel -- C++ around three asm statements -- can be executed on the CPU by the stand-in runtime of this directory.
// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h).
//
// What it models: one wave = 64 lanes with VGPRs v0-v255, AGPRs a0-a255, SGPRs s0-s105, M0, EXEC, VCC, SCC and the VGPR
// index mode (s_set_gpr_idx_on / _idx / _off: M0[7:0] is added to the register number of the selected operand
// positions, AGPR sources included); the register files PERSIST from one asm statement of a kernel to the next (the
// kernel keeps a tree's matrix images in AGPRs across statements).  Code addresses are virtual -- 8 bytes per
// instruction -- which is all the generated code needs: every jump goes through label differences, s_getpc_b64 and
// s_setpc_b64.  LDS addresses are byte offsets into the launch's dynamic LDS (prepare.py rewrites LdsAddress()
// accordingly; M0 carries an 18-bit LDS base for global_load_lds: the images of a 64-taxon tree stage through 126 KB).
// s_waitcnt / s_nop are no-ops: every instruction completes before the next starts.  An LDS read past the launch's
// allocation returns zeros (the image loader over-reads its last rows into registers nothing uses, as the hardware lets
// it); an LDS WRITE out of range aborts.  The ~40 opcodes the generator emits are implemented; an unknown one aborts with
// its text.  Debugging aids (environment): HIP_EMU_ASM_TRACE=<n> prints the first n scalar instructions of a statement,
// HIP_EMU_ASM_NONFINITE=1 the first instructions that write a non-finite double, HIP_EMU_ASM_DIGEST=1 (hip_runtime.h) a
// hash of LDS, operands and register files at every statement's entry and exit -- two runs diffed show where they part.
// v_mfma_f64_4x4x4_4b: four 4 x 4 x 4 blocks; A lane = 16 k + 4 b + i, B lane = 16 k + 4 b + j, D lane = 16 i + 4 b + j
// (bito_amd/csrc/walk_lds.hip), summed as a fused multiply-add chain over k.
#pragma once

#include <cmath>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace hip_emu {

// ---- operands of an asm statement -------------------------------------------------------------------------------
struct AsmOperand {
  const char* name;
  bool scalar, output;
  void* target;   // outputs: where the lane's result goes
  int size;       // bytes (4 or 8)
  uint64_t value; // inputs: the lane's value
};
template <typename T>
inline AsmOperand Op(const char* name, const char* constraint, T& ref) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "asm operands are 32 or 64 bits");
  AsmOperand o{name, std::strchr(constraint, 's') != nullptr, std::strchr(constraint, '=') != nullptr,
               (void*)(&ref), (int)sizeof(T), 0};
  std::memcpy(&o.value, (const void*)(&ref), sizeof(T));
  return o;
}

// ---- the parsed program -------------------------------------------------------------------------------------------
enum class RK : uint8_t { None, V, A, S, Exec, M0, Vcc, Imm, Named };
struct Reg {
  RK kind = RK::None;
  int index = 0, count = 1;  // registers: first index, number of dwords
  uint64_t imm = 0;          // immediates: the value as the assembler would encode it (integer or label difference)
  double fimm = 0;           // ... and as a floating-point literal (1.0, 0.5)
  bool is_float = false;
  int named = -1;            // index into the statement's operand list
};
struct Inst {
  std::string op, text;
  Reg r[4];
  int nr = 0;
  int offset = 0;        // offset:N
  int sel1 = -1;         // SDWA src1_sel BYTE_n
  int idx_mode = 0;      // gpr_idx(...) bits: SRC0 1, SRC1 2, SRC2 4, DST 8
  int target = -1;       // branch target (instruction index)
  int cls = 0;           // InstClass: what the hardware's SQ_INSTS_* counters would file it under
  int wait_lgkm = -1, wait_vm = -1;  // s_waitcnt: the counts it waits for (-1: not named)
};

// Executed wave-level instructions by class, over the process (HIP_EMU_ASM_COUNT=1 prints them at exit as one JSON line on
// stderr).  The same quantities the hardware counts as SQ_INSTS_MFMA / _VALU (matrix instructions included) / _SALU /
// _SMEM / _LDS / _VMEM_RD: profiles/r4_v2_pipe_one_wave_pmc.json holds them for a launch of 6400 DS1 trees, and
// scripts/emu_pipe_instruction_mix.py sets the two side by side -- a check that the interpreter walks the instruction
// stream the device does.  (The C++ around the statements is compiled code on the device and fibers here: not counted.)
enum InstClass { kClassMfma, kClassValu, kClassSalu, kClassSmem, kClassLds, kClassVmemRd, kClassBranch, kClassWait, kClassCount };
struct InstCounts {
  long long n[kClassCount] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long statements = 0;
  ~InstCounts() {
    if (!std::getenv("HIP_EMU_ASM_COUNT")) return;
    std::fprintf(stderr,
                 "{\"asm_statements\": %lld, \"mfma\": %lld, \"valu_other\": %lld, \"salu\": %lld, \"smem\": %lld, \"lds\": %lld, "
                 "\"vmem_rd\": %lld, \"branch\": %lld, \"wait_nop_barrier\": %lld}\n",
                 statements, n[kClassMfma], n[kClassValu], n[kClassSalu], n[kClassSmem], n[kClassLds], n[kClassVmemRd],
                 n[kClassBranch], n[kClassWait]);
  }
};
// Hazards: what the interpreter's one-instruction-at-a-time execution would otherwise hide.  (1) s_waitcnt: a register
// read (or overwritten) while a load that names it may still be in flight, an LDS range read while a global_load_lds may
// still be filling it.  "May": by the ISA's guarantees alone -- lgkmcnt(N) with N > 0 retires the oldest LDS operations
// only as far as the count allows when every scalar load in flight is assumed to have returned first (scalar loads
// return out of order), and never a scalar load.  (2) a scalar write of M0 needs one wait state before s_movrels or
// global_load_lds reads it.  (3) the wait states around v_mfma_f64_4x4x4 that the generator pads by hand inside the
// asm strings -- matrix result -> vector read or overwrite 6, -> matrix A/B operand 6, -> matrix C operand 4, -> address or
// data of an LDS / memory instruction 9; vector result -> matrix operand 2 (scripts/gen_walk_pipe.py's table, measured
// from hipcc's own output for gfx950); every instruction is one state, s_nop n is n + 1, a taken jump counts four more
// (measured there: 21 cycles).  The generator pads statically per body and drains at labels; this check follows the
// path actually executed, across bodies.  Every hazard is printed (the first sixteen) and counted; HIP_EMU_ASM_HAZARDS=abort makes
// the first one fatal, =0 switches the check off.  The count is printed at exit with HIP_EMU_ASM_COUNT=1.
struct HazardLog {
  long long count = 0, checked_reads = 0, waits = 0;
  int mode = 1;  // 0 off, 1 report, 2 abort
  HazardLog() {
    if (const char* m = std::getenv("HIP_EMU_ASM_HAZARDS")) mode = std::strcmp(m, "abort") == 0 ? 2 : std::atoi(m);
  }
  ~HazardLog() {
    if (std::getenv("HIP_EMU_ASM_COUNT"))
      std::fprintf(stderr, "{\"asm_hazards\": %lld, \"register_reads_checked\": %lld, \"s_waitcnt_executed\": %lld}\n", count, checked_reads, waits);
  }
  void Report(const char* what, const char* inst, const char* producer) {
    if (__atomic_fetch_add(&count, 1, __ATOMIC_RELAXED) < 16) std::fprintf(stderr, "gfx950_asm: HAZARD: %s: %s   (in flight: %s)\n", what, inst, producer);
    if (mode == 2) std::abort();
  }
};
inline HazardLog& Hazards() {
  static HazardLog h;
  return h;
}
inline InstCounts& Counts() {
  static InstCounts c;
  return c;
}
inline int ClassOf(const std::string& op) {
  if (op.rfind("v_mfma", 0) == 0) return kClassMfma;
  if (op.rfind("v_", 0) == 0) return kClassValu;
  if (op.rfind("ds_", 0) == 0) return kClassLds;
  if (op.rfind("global_", 0) == 0 || op.rfind("buffer_", 0) == 0) return kClassVmemRd;
  if (op.rfind("s_load", 0) == 0) return kClassSmem;
  if (op == "s_nop" || op == "s_waitcnt" || op == "s_barrier") return kClassWait;
  if (op == "s_branch" || op.rfind("s_cbranch", 0) == 0 || op == "s_setpc_b64") return kClassBranch;
  return kClassSalu;
}
struct Program {
  std::vector<Inst> code;
  std::vector<std::string> operand_names;
};

inline uint64_t ParseInt(const std::string& t) { return (uint64_t)std::strtoll(t.c_str(), nullptr, 0); }

inline Program ParseProgram(const char* text) {
  Program P;
  std::vector<std::string> lines;
  {
    std::string cur;
    for (const char* c = text; *c; c++) {
      if (*c == '\n') { lines.push_back(cur); cur.clear(); }
      else cur.push_back(*c);
    }
    if (!cur.empty()) lines.push_back(cur);
  }
  std::map<std::string, int> labels;
  std::vector<std::pair<int, std::string>> body;  // (instruction index, text)
  for (std::string ln : lines) {
    const size_t semi = ln.find(';');
    if (semi != std::string::npos) ln = ln.substr(0, semi);
    size_t a = ln.find_first_not_of(" \t"), b = ln.find_last_not_of(" \t");
    if (a == std::string::npos) continue;
    ln = ln.substr(a, b - a + 1);
    if (ln.back() == ':') { labels[ln.substr(0, ln.size() - 1)] = (int)body.size(); continue; }
    body.push_back({(int)body.size(), ln});
  }
  auto operand_index = [&](const std::string& name) {
    for (size_t k = 0; k < P.operand_names.size(); k++)
      if (P.operand_names[k] == name) return (int)k;
    P.operand_names.push_back(name);
    return (int)P.operand_names.size() - 1;
  };
  auto label_address = [&](const std::string& name) -> int64_t {
    auto it = labels.find(name);
    if (it == labels.end()) { std::fprintf(stderr, "gfx950_asm: unknown label %s\n", name.c_str()); std::abort(); }
    return (int64_t)it->second * 8;
  };
  auto parse_reg = [&](std::string t) -> Reg {
    Reg r;
    if (t.rfind("%[", 0) == 0) { r.kind = RK::Named; r.named = operand_index(t.substr(2, t.size() - 3)); return r; }
    if (t == "exec") { r.kind = RK::Exec; r.count = 2; return r; }
    if (t == "vcc") { r.kind = RK::Vcc; r.count = 2; return r; }
    if (t == "m0") { r.kind = RK::M0; return r; }
    if ((t[0] == 'v' || t[0] == 'a' || t[0] == 's') && t.size() > 1 && (std::isdigit((unsigned char)t[1]) || t[1] == '[')) {
      r.kind = t[0] == 'v' ? RK::V : (t[0] == 'a' ? RK::A : RK::S);
      if (t[1] == '[') {
        const size_t colon = t.find(':');
        r.index = std::atoi(t.substr(2, colon - 2).c_str());
        r.count = std::atoi(t.substr(colon + 1).c_str()) - r.index + 1;
      } else {
        r.index = std::atoi(t.c_str() + 1);
      }
      return r;
    }
    r.kind = RK::Imm;
    if (t[0] == '.') {  // label, or label - label
      const size_t minus = t.find('-');
      r.imm = minus == std::string::npos ? (uint64_t)label_address(t)
                                         : (uint64_t)(label_address(t.substr(0, minus)) - label_address(t.substr(minus + 1)));
      return r;
    }
    if (t.find('.') != std::string::npos && t.find("0x") == std::string::npos) {
      r.is_float = true;
      r.fimm = std::atof(t.c_str());
      return r;
    }
    r.imm = ParseInt(t);
    r.fimm = (double)(int64_t)r.imm;  // (an integer inline constant used by a floating-point instruction: 0)
    return r;
  };
  for (auto& [index, ln] : body) {
    Inst in;
    in.text = ln;
    const size_t sp = ln.find_first_of(" \t");
    in.op = ln.substr(0, sp);
    std::string rest = sp == std::string::npos ? "" : ln.substr(sp + 1);
    // modifiers
    auto take = [&](const char* key) -> std::string {
      const size_t at = rest.find(key);
      if (at == std::string::npos) return "";
      size_t end = rest.find_first_of(" \t", at);
      std::string v = rest.substr(at + std::strlen(key), end == std::string::npos ? std::string::npos : end - at - std::strlen(key));
      rest.erase(at, end == std::string::npos ? std::string::npos : end - at);
      return v;
    };
    std::string v;
    if (!(v = take("offset:")).empty()) in.offset = (int)ParseInt(v);
    if (!(v = take("src1_sel:")).empty()) in.sel1 = v == "DWORD" ? -1 : v.back() - '0';
    take("dst_sel:"); take("dst_unused:"); take("src0_sel:");
    {
      const size_t g = rest.find("gpr_idx(");
      if (g != std::string::npos) {
        const std::string list = rest.substr(g + 8, rest.find(')', g) - g - 8);
        if (list.find("SRC0") != std::string::npos) in.idx_mode |= 1;
        if (list.find("SRC1") != std::string::npos) in.idx_mode |= 2;
        if (list.find("SRC2") != std::string::npos) in.idx_mode |= 4;
        if (list.find("DST") != std::string::npos) in.idx_mode |= 8;
        rest.erase(g);
      }
    }
    if (in.op == "s_waitcnt") {
      const size_t l = rest.find("lgkmcnt("), m = rest.find("vmcnt(");
      if (l != std::string::npos) in.wait_lgkm = std::atoi(rest.c_str() + l + 8);
      if (m != std::string::npos) in.wait_vm = std::atoi(rest.c_str() + m + 6);
      if (rest.find("expcnt(") != std::string::npos) { std::fprintf(stderr, "gfx950_asm: expcnt is not modelled: %s\n", ln.c_str()); std::abort(); }
    }
    if (in.op == "s_nop") in.offset = std::atoi(rest.c_str());
    if (in.op == "s_waitcnt" || in.op == "s_nop" || in.op == "s_barrier" || in.op == "s_set_gpr_idx_off") rest.clear();
    // operands
    size_t at = 0;
    while (at < rest.size()) {
      size_t comma = rest.find(',', at);
      // (a register range has no comma inside; label differences neither)
      std::string tok = rest.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
      const size_t x = tok.find_first_not_of(" \t"), y = tok.find_last_not_of(" \t");
      if (x != std::string::npos) {
        tok = tok.substr(x, y - x + 1);
        if (in.op == "s_branch" || in.op.rfind("s_cbranch", 0) == 0) in.target = (int)(label_address(tok) / 8);
        else if (in.nr < 4) in.r[in.nr++] = parse_reg(tok);
      }
      if (comma == std::string::npos) break;
      at = comma + 1;
    }
    in.cls = ClassOf(in.op);
    P.code.push_back(in);
  }
  return P;
}

// ---- the machine ----------------------------------------------------------------------------------------------------
struct WaveMachine {
  std::vector<uint32_t> v, a;  // [register][lane]
  uint32_t s[128];
  uint32_t m0 = 0;
  uint64_t exec = ~0ull, vcc = 0;
  bool scc = false, idx_on = false;
  int idx_bits = 0;
  // Memory operations in flight, as s_waitcnt sees them (the VALUES are delivered at once -- this is bookkeeping for the
  // hazard check below).  lgkm: LDS operations (complete in order among themselves) and scalar loads (complete in any
  // order); vm: global_load_lds (in order).  A register is "pending" from the load that names it to the s_waitcnt that
  // guarantees its arrival; so is the LDS range a global_load_lds fills.
  struct InFlight {
    bool scalar = false;            // s_load*
    std::vector<int> regs;          // destination registers: file << 16 | index (file 0 v, 1 a, 2 s)
    uint32_t lds_lo = 0, lds_hi = 0;  // LDS bytes a global_load_lds writes
    const char* text = "";
  };
  std::deque<InFlight> lgkm, vm;
  std::vector<uint16_t> pending[3];  // per register of each file: loads in flight that will write it
  long m0_age = 1000;                // wait states since a scalar instruction wrote M0
  // wait states between a producer and a dependent instruction (the table scripts/gen_walk_pipe.py works from, measured
  // from what hipcc inserts for gfx950 around v_mfma_f64_4x4x4): where each VGPR / AGPR was last written, and by what
  long issue_pos = 0;                        // wait states issued so far in this statement
  std::vector<long> written_at[2];           // per VGPR / AGPR: issue_pos of its last write (very negative: long ago)
  std::vector<uint8_t> written_by[2];        // 1 a matrix instruction, 2 another vector instruction, 0 anything else
  WaveMachine() : v((size_t)384 * 64, 0), a((size_t)256 * 64, 0) {
    std::memset(s, 0, sizeof(s));
    pending[0].assign(384, 0);
    pending[1].assign(256, 0);
    pending[2].assign(128, 0);
    written_at[0].assign(384, -1000);
    written_at[1].assign(256, -1000);
    written_by[0].assign(384, 0);
    written_by[1].assign(256, 0);
  }
  uint32_t& V(int r, int lane) {
    if (r < 0 || r >= 384) { std::fprintf(stderr, "gfx950_asm: VGPR v%d does not exist\n", r); std::abort(); }
    return v[(size_t)r * 64 + lane];
  }
  uint32_t& A(int r, int lane) {
    if (r < 0 || r >= 256) { std::fprintf(stderr, "gfx950_asm: AGPR a%d does not exist\n", r); std::abort(); }
    return a[(size_t)r * 64 + lane];
  }
};

struct AsmContext {
  char* lds;                      // the launch's dynamic LDS
  size_t lds_bytes;
  std::function<void()> barrier;  // s_barrier: every wave of the workgroup
};

[[noreturn]] inline void AsmFail(const Inst& in, const char* why) {
  std::fprintf(stderr, "gfx950_asm: %s: %s\n", why, in.text.c_str());
  std::abort();
}

// Runs one asm statement for one wave.  operands[k][lane] = the value lane `lane` bound to operand k of the program
// (scalar operands: lane 0's); outputs are written back into the same array.
inline void Execute(const Program& P, WaveMachine& M, std::vector<std::vector<uint64_t>>& operands,
                    const std::vector<bool>& scalar, const AsmContext& ctx) {
  // named operands live in registers of their own: SGPR pairs s[106 + 2k], VGPR pairs v[256 + 2k]
  auto named_s = [&](int k) { return 106 + 2 * k; };
  auto named_v = [&](int k) { return 256 + 2 * k; };
  for (size_t k = 0; k < operands.size(); k++) {
    if (scalar[k]) {
      if (named_s((int)k) + 1 >= 128) { std::fprintf(stderr, "gfx950_asm: too many scalar operands\n"); std::abort(); }
      M.s[named_s((int)k)] = (uint32_t)operands[k][0];
      M.s[named_s((int)k) + 1] = (uint32_t)(operands[k][0] >> 32);
    } else {
      for (int l = 0; l < 64; l++) {
        M.V(named_v((int)k), l) = (uint32_t)operands[k][l];
        M.V(named_v((int)k) + 1, l) = (uint32_t)(operands[k][l] >> 32);
      }
    }
  }
  auto resolve = [&](const Reg& r) -> Reg {  // named operand -> its register
    if (r.kind != RK::Named) return r;
    Reg o;
    o.kind = scalar[(size_t)r.named] ? RK::S : RK::V;
    o.index = scalar[(size_t)r.named] ? named_s(r.named) : named_v(r.named);
    o.count = 2;
    return o;
  };
  // position: 0 = dst, 1..3 = src0..src2 (index mode)
  // (the index mode acts on vector ALU and matrix instructions only: memory instructions pass position 9)
  auto vindex = [&](const Reg& r, int position) {
    int idx = r.index;
    if (M.idx_on && position < 4 && (r.kind == RK::V || r.kind == RK::A) && r.index < 256) {
      const int bit = position == 0 ? 8 : (1 << (position - 1));
      if (M.idx_bits & bit) idx += (int)(M.m0 & 0xff);
    }
    return idx;
  };
  HazardLog& hazards = Hazards();
  const Inst* current = nullptr;
  int check_lane = 0;  // the first lane of EXEC: register checks are made once per instruction, not per lane
  int inst_kind = 0;  // of the instruction being executed: 1 matrix, 2 other vector, 3 LDS / memory, 0 scalar
  auto spacing = [&](int file, int index, int position) {  // wait states behind the register's producer (files 0, 1)
    const int by = M.written_by[file][(size_t)index];
    if (!by || !inst_kind) return;
    int need = 0;
    if (by == 1) need = inst_kind == 1 ? (position == 3 ? 4 : position == 0 ? 0 : 6) : inst_kind == 2 ? 6 : 9;
    else if (by == 2 && inst_kind == 1 && position != 0) need = 2;
    const long have = M.issue_pos - M.written_at[file][(size_t)index] - 1;
    if (have < need) {
      char what[160];
      std::snprintf(what, sizeof(what), "%ld wait states behind a %s result where %d are needed (%s%d as operand %d)", have,
                    by == 1 ? "matrix" : "vector", need, file ? "a" : "v", index, position);
      hazards.Report(what, current ? current->text.c_str() : "", "");
    }
  };
  auto touch = [&](int file, int index, const char* what) {  // a register about to be read or overwritten
    __atomic_fetch_add(&hazards.checked_reads, 1, __ATOMIC_RELAXED);
    if (!M.pending[file][(size_t)index]) return;
    const char* producer = "";
    for (const auto* q : {&M.lgkm, &M.vm})
      for (const WaveMachine::InFlight& f : *q)
        for (int reg : f.regs)
          if (reg == (file << 16 | index)) producer = f.text;
    hazards.Report(what, current ? current->text.c_str() : "", producer);
  };
  auto read32 = [&](const Reg& r0, int position, int lane, int word = 0) -> uint32_t {
    const Reg r = resolve(r0);
    switch (r.kind) {
      case RK::V: {
        const int idx = vindex(r, position) + word;
        if (hazards.mode && (lane == 0 || lane == check_lane)) {
          touch(0, idx, "reads a VGPR before the s_waitcnt that delivers it");
          spacing(0, idx, position);
        }
        return M.V(idx, lane);
      }
      case RK::A: {
        const int idx = vindex(r, position) + word;
        if (hazards.mode && (lane == 0 || lane == check_lane)) {
          touch(1, idx, "reads an AGPR before the s_waitcnt that delivers it");
          spacing(1, idx, position);
        }
        return M.A(idx, lane);
      }
      case RK::S:
        if (hazards.mode) touch(2, r.index + word, "reads an SGPR before the s_waitcnt that delivers it");
        return M.s[r.index + word];
      case RK::M0: return M.m0;
      case RK::Exec: return (uint32_t)(M.exec >> (32 * word));
      case RK::Vcc: return (uint32_t)(M.vcc >> (32 * word));
      case RK::Imm: return (uint32_t)(r.imm >> (32 * word));
      default: return 0;
    }
  };
  auto read64 = [&](const Reg& r, int position, int lane) -> uint64_t {
    const Reg q = resolve(r);
    if (q.kind == RK::Imm) return q.imm;  // (integer immediates are sign-extended by ParseInt)
    return (uint64_t)read32(r, position, lane, 0) | ((uint64_t)read32(r, position, lane, 1) << 32);
  };
  auto readf64 = [&](const Reg& r, int position, int lane) -> double {
    const Reg q = resolve(r);
    if (q.kind == RK::Imm) return q.is_float ? q.fimm : (double)(int64_t)q.imm;
    const uint64_t bits = read64(r, position, lane);
    double d;
    std::memcpy(&d, &bits, 8);
    return d;
  };
  auto write32 = [&](const Reg& r0, int lane, uint32_t value, int word = 0) {
    const Reg r = resolve(r0);
    switch (r.kind) {
      case RK::V:
        if (hazards.mode && (lane == 0 || lane == check_lane)) {
          touch(0, vindex(r, 0) + word, "overwrites a VGPR a load in flight will write");
          if (inst_kind == 2) spacing(0, vindex(r, 0) + word, 9);  // (a vector instruction over a matrix result: as a read)
          M.written_at[0][(size_t)(vindex(r, 0) + word)] = M.issue_pos;
          M.written_by[0][(size_t)(vindex(r, 0) + word)] = (uint8_t)(inst_kind <= 2 ? inst_kind : 0);
        }
        M.V(vindex(r, 0) + word, lane) = value;
        break;
      case RK::A:
        if (hazards.mode && (lane == 0 || lane == check_lane)) {
          touch(1, vindex(r, 0) + word, "overwrites an AGPR a load in flight will write");
          if (inst_kind == 2) spacing(1, vindex(r, 0) + word, 9);
          M.written_at[1][(size_t)(vindex(r, 0) + word)] = M.issue_pos;
          M.written_by[1][(size_t)(vindex(r, 0) + word)] = (uint8_t)(inst_kind <= 2 ? inst_kind : 0);
        }
        M.A(vindex(r, 0) + word, lane) = value;
        break;
      case RK::S:
        if (hazards.mode) touch(2, r.index + word, "overwrites an SGPR a load in flight will write");
        M.s[r.index + word] = value;
        break;
      case RK::M0:
        M.m0 = value;
        M.m0_age = -1;  // (the instruction's own wait state is added at the end of the loop body)
        break;
      case RK::Exec: M.exec = word ? ((M.exec & 0xffffffffull) | ((uint64_t)value << 32)) : ((M.exec & ~0xffffffffull) | value); break;
      default: break;
    }
  };
  auto write_raw = [&](const Reg& r0, int lane, uint32_t value, int word) {  // a memory instruction's destination
    const Reg r = resolve(r0);
    if (r.kind == RK::V) {
      M.V(r.index + word, lane) = value;
      M.written_by[0][(size_t)(r.index + word)] = 0;
    } else if (r.kind == RK::A) {
      M.A(r.index + word, lane) = value;
      M.written_by[1][(size_t)(r.index + word)] = 0;
    }
  };
  static const bool report_nonfinite = std::getenv("HIP_EMU_ASM_NONFINITE") != nullptr;
  static int nonfinite_reports = 0;
  auto writef64 = [&](const Reg& r, int lane, double d) {
    if (report_nonfinite && !std::isfinite(d) && nonfinite_reports < 8) {
      nonfinite_reports++;
      std::fprintf(stderr, "gfx950_asm: lane %d gets %g from: %s (m0 %u, index mode %d bits %d)\n", lane, d, current ? current->text.c_str() : "", M.m0,
                   (int)M.idx_on, M.idx_bits);
    }
    uint64_t bits;
    std::memcpy(&bits, &d, 8);
    write32(r, lane, (uint32_t)bits, 0);
    write32(r, lane, (uint32_t)(bits >> 32), 1);
  };
  auto active = [&](int lane) { return (M.exec >> lane) & 1; };
  auto sreg64 = [&](const Reg& r0) -> uint64_t {
    const Reg r = resolve(r0);
    if (r.kind == RK::Imm) return r.imm;
    return (uint64_t)read32(r, 1, 0, 0) | ((uint64_t)read32(r, 1, 0, 1) << 32);
  };

  auto lds_at = [&](uint32_t addr, size_t bytes) -> char* {
    if ((size_t)addr + bytes > ctx.lds_bytes) {
      std::fprintf(stderr, "gfx950_asm: LDS access at %u (+%zu) beyond the launch's %zu bytes: %s\n", addr, bytes, ctx.lds_bytes,
                   current ? current->text.c_str() : "");
      std::abort();
    }
    return ctx.lds + addr;
  };
  auto issue = [&](std::deque<WaveMachine::InFlight>& queue, bool scalar_load, const Reg* dst, int words, const Inst& in) -> WaveMachine::InFlight& {
    queue.emplace_back();
    WaveMachine::InFlight& f = queue.back();
    f.scalar = scalar_load;
    f.text = in.text.c_str();
    if (dst) {
      const Reg r = resolve(*dst);
      const int file = r.kind == RK::V ? 0 : r.kind == RK::A ? 1 : 2;
      for (int w = 0; w < words; w++) {
        f.regs.push_back(file << 16 | (r.index + w));
        M.pending[file][(size_t)(r.index + w)]++;
      }
    }
    return f;
  };
  auto retire = [&](std::deque<WaveMachine::InFlight>& queue, size_t position) {
    for (int reg : queue[position].regs) M.pending[reg >> 16][(size_t)(reg & 0xffff)]--;
    queue.erase(queue.begin() + (long)position);
  };
  // s_waitcnt lgkmcnt(n) / vmcnt(n): what the ISA guarantees has arrived (see HazardLog)
  auto wait_for = [&](int lgkm, int vm) {
    __atomic_fetch_add(&hazards.waits, 1, __ATOMIC_RELAXED);
    if (lgkm == 0) {
      while (!M.lgkm.empty()) retire(M.lgkm, 0);
    } else if (lgkm > 0) {
      size_t scalar_loads = 0;
      for (const auto& f : M.lgkm) scalar_loads += f.scalar;
      const long completions = (long)M.lgkm.size() - lgkm;       // at least this many have returned ...
      long lds_done = completions - (long)scalar_loads;           // ... the scalar loads first, for all we know
      for (size_t k = 0; k < M.lgkm.size() && lds_done > 0;) {
        if (M.lgkm[k].scalar) { k++; continue; }
        retire(M.lgkm, k);
        lds_done--;
      }
    }
    if (vm >= 0)
      while ((long)M.vm.size() > vm) retire(M.vm, 0);
  };
  size_t pc = 0;
  long executed = 0;
  __atomic_fetch_add(&Counts().statements, 1, __ATOMIC_RELAXED);
  M.issue_pos += 1000;  // (compiled code stands between two statements: their producers are long ago)
  while (pc < P.code.size()) {
    const Inst& in = P.code[pc];
    current = &in;
    const std::string& op = in.op;
    size_t next = pc + 1;
    check_lane = M.exec ? __builtin_ctzll(M.exec) : 0;
    inst_kind = in.cls == kClassMfma ? 1 : in.cls == kClassValu ? 2 : (in.cls == kClassLds || in.cls == kClassVmemRd) ? 3 : 0;
    if (++executed > 50000000) AsmFail(in, "no end in sight");
    __atomic_fetch_add(&Counts().n[in.cls], 1, __ATOMIC_RELAXED);
    static const long trace_until = std::getenv("HIP_EMU_ASM_TRACE") ? std::atol(std::getenv("HIP_EMU_ASM_TRACE")) : 0;
    if (executed <= trace_until && (op[0] == 's' && op != "s_nop" && op != "s_waitcnt"))
      std::fprintf(stderr, "  [%ld] pc %zu  %s   (m0 %u scc %d s32 %08x s48 %08x s92 %u)\n", executed, pc, in.text.c_str(), M.m0, (int)M.scc,
                   M.s[32], M.s[48], M.s[92]);
    if (op == "s_nop") {
      M.m0_age += in.offset;  // (s_nop n: n + 1 wait states, the one every instruction adds below included)
    } else if (op == "s_waitcnt") {
      if (hazards.mode) wait_for(in.wait_lgkm, in.wait_vm);
    } else if (op == "s_barrier") {
      ctx.barrier();
    } else if (op == "s_mov_b32") {
      write32(in.r[0], 0, read32(in.r[1], 1, 0));
    } else if (op == "s_mov_b64") {
      const uint64_t val = sreg64(in.r[1]);
      const Reg d = resolve(in.r[0]);
      if (d.kind == RK::Exec) M.exec = val;
      else { write32(in.r[0], 0, (uint32_t)val, 0); write32(in.r[0], 0, (uint32_t)(val >> 32), 1); }
    } else if (op == "s_add_u32" || op == "s_addc_u32") {
      const uint64_t sum = (uint64_t)read32(in.r[1], 1, 0) + read32(in.r[2], 2, 0) + (op == "s_addc_u32" && M.scc ? 1 : 0);
      write32(in.r[0], 0, (uint32_t)sum);
      M.scc = (sum >> 32) != 0;
    } else if (op == "s_lshl_b32") {
      const uint32_t res = read32(in.r[1], 1, 0) << (read32(in.r[2], 2, 0) & 31);
      write32(in.r[0], 0, res);
      M.scc = res != 0;
    } else if (op == "s_lshr_b32") {
      const uint32_t res = read32(in.r[1], 1, 0) >> (read32(in.r[2], 2, 0) & 31);
      write32(in.r[0], 0, res);
      M.scc = res != 0;
    } else if (op == "s_and_b32") {
      const uint32_t res = read32(in.r[1], 1, 0) & read32(in.r[2], 2, 0);
      write32(in.r[0], 0, res);
      M.scc = res != 0;
    } else if (op == "s_bfe_u32") {
      const uint32_t src = read32(in.r[1], 1, 0), spec = read32(in.r[2], 2, 0);
      const uint32_t off = spec & 31, width = (spec >> 16) & 0x7f;
      const uint32_t res = width == 0 ? 0 : (width >= 32 ? src >> off : ((src >> off) & ((1u << width) - 1)));
      write32(in.r[0], 0, res);
      M.scc = res != 0;
    } else if (op == "s_bitcmp1_b32") {
      M.scc = (read32(in.r[0], 1, 0) >> (read32(in.r[1], 2, 0) & 31)) & 1;
    } else if (op == "s_cselect_b64") {
      const uint64_t val = M.scc ? sreg64(in.r[1]) : sreg64(in.r[2]);
      const Reg d = resolve(in.r[0]);
      if (d.kind == RK::Exec) M.exec = val;
      else { write32(in.r[0], 0, (uint32_t)val, 0); write32(in.r[0], 0, (uint32_t)(val >> 32), 1); }
    } else if (op == "s_cmp_le_u32") {
      M.scc = read32(in.r[0], 1, 0) <= read32(in.r[1], 2, 0);
    } else if (op == "s_cmp_ge_u32") {
      M.scc = read32(in.r[0], 1, 0) >= read32(in.r[1], 2, 0);
    } else if (op == "s_cmp_lt_u32") {
      M.scc = read32(in.r[0], 1, 0) < read32(in.r[1], 2, 0);
    } else if (op == "s_cmp_eq_u32") {
      M.scc = read32(in.r[0], 1, 0) == read32(in.r[1], 2, 0);
    } else if (op == "s_cmp_lg_u32") {
      M.scc = read32(in.r[0], 1, 0) != read32(in.r[1], 2, 0);
    } else if (op == "s_branch") {
      next = (size_t)in.target;
    } else if (op == "s_cbranch_scc1") {
      if (M.scc) next = (size_t)in.target;
    } else if (op == "s_cbranch_scc0") {
      if (!M.scc) next = (size_t)in.target;
    } else if (op == "s_getpc_b64") {
      const uint64_t addr = (uint64_t)(pc + 1) * 8;
      write32(in.r[0], 0, (uint32_t)addr, 0);
      write32(in.r[0], 0, (uint32_t)(addr >> 32), 1);
    } else if (op == "s_setpc_b64") {
      const uint64_t addr = sreg64(in.r[0]);
      if (addr % 8 || addr / 8 > P.code.size()) AsmFail(in, "a jump outside the statement");
      next = (size_t)(addr / 8);
    } else if (op == "s_movrels_b32") {
      if (hazards.mode && M.m0_age < 1) hazards.Report("s_movrels reads M0 without a wait state behind the scalar write", in.text.c_str(), "m0");
      const Reg src = resolve(in.r[1]);
      write32(in.r[0], 0, M.s[(src.index + (int)M.m0) & 127]);
    } else if (op == "s_set_gpr_idx_on") {
      M.m0 = (M.m0 & ~0xffu) | (read32(in.r[0], 1, 0) & 0xff);
      M.idx_bits = in.idx_mode;
      M.idx_on = true;
    } else if (op == "s_set_gpr_idx_idx") {
      M.m0 = (M.m0 & ~0xffu) | (read32(in.r[0], 1, 0) & 0xff);
    } else if (op == "s_set_gpr_idx_off") {
      M.idx_on = false;
    } else if (op == "s_load_dword" || op == "s_load_dwordx16" || op == "s_load_dwordx8" || op == "s_load_dwordx4" ||
               op == "s_load_dwordx2") {
      const int words = op == "s_load_dword" ? 1 : std::atoi(op.c_str() + 13);
      const uint64_t base = sreg64(in.r[1]);
      const uint64_t off = in.nr > 2 ? (resolve(in.r[2]).kind == RK::Imm ? in.r[2].imm : read32(in.r[2], 2, 0)) : 0;
      const Reg d = resolve(in.r[0]);
      // (a load over a load in flight is not flagged: the loader warms the scalar cache with loads into one scratch
      // register, and LDS reads return in order)
      std::memcpy(&M.s[d.index], reinterpret_cast<const void*>(base + off), (size_t)words * 4);
      if (hazards.mode) issue(M.lgkm, true, &in.r[0], words, in);
    } else if (op == "v_mov_b32") {
      for (int l = 0; l < 64; l++)
        if (active(l)) write32(in.r[0], l, read32(in.r[1], 1, l));
    } else if (op == "v_mov_b64") {
      for (int l = 0; l < 64; l++)
        if (active(l)) {
          const Reg q = resolve(in.r[1]);
          if (q.kind == RK::Imm) {
            if (q.is_float) writef64(in.r[0], l, q.fimm);
            else { write32(in.r[0], l, (uint32_t)q.imm, 0); write32(in.r[0], l, (uint32_t)(q.imm >> 32), 1); }
          } else {
            const uint32_t lo = read32(in.r[1], 1, l, 0), hi = read32(in.r[1], 1, l, 1);
            write32(in.r[0], l, lo, 0);
            write32(in.r[0], l, hi, 1);
          }
        }
    } else if (op == "v_add_u32") {
      for (int l = 0; l < 64; l++)
        if (active(l)) write32(in.r[0], l, read32(in.r[1], 1, l) + read32(in.r[2], 2, l));
    } else if (op == "v_sub_u32") {
      for (int l = 0; l < 64; l++)
        if (active(l)) write32(in.r[0], l, read32(in.r[1], 1, l) - read32(in.r[2], 2, l));
    } else if (op == "v_and_b32") {
      for (int l = 0; l < 64; l++)
        if (active(l)) write32(in.r[0], l, read32(in.r[1], 1, l) & read32(in.r[2], 2, l));
    } else if (op == "v_lshlrev_b32") {
      for (int l = 0; l < 64; l++)
        if (active(l)) write32(in.r[0], l, read32(in.r[2], 2, l) << (read32(in.r[1], 1, l) & 31));
    } else if (op == "v_lshlrev_b32_sdwa") {
      for (int l = 0; l < 64; l++)
        if (active(l)) {
          uint32_t src1 = read32(in.r[2], 2, l);
          if (in.sel1 >= 0) src1 = (src1 >> (8 * in.sel1)) & 0xff;
          write32(in.r[0], l, src1 << (read32(in.r[1], 1, l) & 31));
        }
    } else if (op == "v_mul_f64") {
      for (int l = 0; l < 64; l++)
        if (active(l)) writef64(in.r[0], l, readf64(in.r[1], 1, l) * readf64(in.r[2], 2, l));
    } else if (op == "v_add_f64") {
      for (int l = 0; l < 64; l++)
        if (active(l)) writef64(in.r[0], l, readf64(in.r[1], 1, l) + readf64(in.r[2], 2, l));
    } else if (op == "v_fma_f64") {
      for (int l = 0; l < 64; l++)
        if (active(l)) writef64(in.r[0], l, std::fma(readf64(in.r[1], 1, l), readf64(in.r[2], 2, l), readf64(in.r[3], 3, l)));
    } else if (op == "v_mfma_f64_4x4x4_4b_f64") {
      double A[64], B[64], C[64], D[64];
      for (int l = 0; l < 64; l++) {
        A[l] = readf64(in.r[1], 1, l);
        B[l] = readf64(in.r[2], 2, l);
        C[l] = readf64(in.r[3], 3, l);
      }
      for (int i = 0; i < 4; i++)
        for (int b = 0; b < 4; b++)
          for (int j = 0; j < 4; j++) {
            double acc = C[16 * i + 4 * b + j];
            for (int k = 0; k < 4; k++) acc = std::fma(A[16 * k + 4 * b + i], B[16 * k + 4 * b + j], acc);
            D[16 * i + 4 * b + j] = acc;
          }
      for (int l = 0; l < 64; l++) writef64(in.r[0], l, D[l]);  // (the matrix instruction ignores EXEC)
    } else if (op == "ds_read_b64" || op == "ds_read_b128") {
      const int words = op == "ds_read_b64" ? 2 : 4;
      for (int l = 0; l < 64; l++)
        if (active(l)) {
          const uint32_t addr = read32(in.r[1], 9, l) + (uint32_t)in.offset;
          if (hazards.mode)
            for (const WaveMachine::InFlight& f : M.vm)
              if (addr < f.lds_hi && addr + (uint32_t)words * 4 > f.lds_lo)
                hazards.Report("reads LDS bytes a global_load_lds may still be filling (no vmcnt wait)", in.text.c_str(), f.text);
          uint32_t tmp[4];
          // (a read past the launch's allocation returns zeros, as the hardware's out-of-range LDS reads do: the image
          // loader's last rows over-read into registers nothing uses; a WRITE out of range stays an error)
          if ((size_t)addr + (size_t)words * 4 > ctx.lds_bytes) std::memset(tmp, 0, sizeof(tmp));
          else std::memcpy(tmp, lds_at(addr, (size_t)words * 4), (size_t)words * 4);
          for (int w = 0; w < words; w++) write_raw(in.r[0], l, tmp[w], w);
        }
      if (hazards.mode) issue(M.lgkm, false, &in.r[0], words, in);
    } else if (op == "ds_write_b64" || op == "ds_write_b128") {
      if (hazards.mode) issue(M.lgkm, false, nullptr, 0, in);
      const int words = op == "ds_write_b64" ? 2 : 4;
      for (int l = 0; l < 64; l++)
        if (active(l)) {
          const uint32_t addr = read32(in.r[0], 9, l) + (uint32_t)in.offset;
          uint32_t tmp[4];
          for (int w = 0; w < words; w++) tmp[w] = read32(in.r[1], 9, l, w);
          std::memcpy(lds_at(addr, (size_t)words * 4), tmp, (size_t)words * 4);
        }
    } else if (op == "ds_add_f64") {
      if (hazards.mode) issue(M.lgkm, false, nullptr, 0, in);
      for (int l = 0; l < 64; l++)
        if (active(l)) {
          const uint32_t addr = read32(in.r[0], 9, l) + (uint32_t)in.offset;
          double cur;
          std::memcpy(&cur, lds_at(addr, 8), 8);
          cur += readf64(in.r[1], 9, l);
          std::memcpy(lds_at(addr, 8), &cur, 8);
        }
    } else if (op == "global_load_lds_dwordx4") {
      // every lane: 16 bytes from (scalar base + its vector offset + offset) to LDS at M0 + offset + 16 * lane
      const uint64_t base = sreg64(in.r[1]);
      if (hazards.mode) {
        if (M.m0_age < 1) hazards.Report("global_load_lds reads M0 without a wait state behind the scalar write", in.text.c_str(), "m0");
        WaveMachine::InFlight& f = issue(M.vm, false, nullptr, 0, in);
        f.lds_lo = (M.m0 & 0x3ffff) + (uint32_t)in.offset;
        f.lds_hi = f.lds_lo + 1024;
      }
      for (int l = 0; l < 64; l++)
        if (active(l)) {
          const uint64_t from = base + read32(in.r[0], 9, l) + (uint64_t)in.offset;
          std::memcpy(lds_at((M.m0 & 0x3ffff) + (uint32_t)in.offset + 16u * (uint32_t)l, 16), reinterpret_cast<const void*>(from), 16);
        }
    } else {
      AsmFail(in, "opcode not implemented");
    }
    M.m0_age++;
    M.issue_pos += 1 + (op == "s_nop" ? in.offset : 0) + (next != pc + 1 ? 4 : 0);  // (a taken jump: four more states)
    pc = next;
  }
  for (size_t k = 0; k < operands.size(); k++) {
    if (scalar[k]) continue;
    for (int l = 0; l < 64; l++)
      operands[k][l] = (uint64_t)M.V(named_v((int)k), l) | ((uint64_t)M.V(named_v((int)k) + 1, l) << 32);
  }
}

}  // namespace hip_emu
