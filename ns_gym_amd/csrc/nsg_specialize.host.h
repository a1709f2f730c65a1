// nsg_specialize.host.h — config-specialised code objects (host side only).
//
// The generic kernels read the wrapper configuration (nsg_config: scheduler / update-fn kinds, θ slots,
// flags, TimeLimit …) through scalar loads and branch on it; for a batch whose configuration is fixed for
// millions of steps that is ~45 % of the scalar + vector instructions of a CartPole step.  nsg_specialize()
// re-compiles THE SAME kernel bodies (step_body / rollout_body, nsg_kernels.hip.h / nsg_rollout.hip.h)
// with the handle's nsg_config as a compile-time constant, through hiprtc, into a gfx950 code object:
// the per-param loop unrolls, every kind switch folds, the config loads disappear.  Results are
// bit-identical to the generic kernels (same source, same -ffp-contract=off), tests/test_gpu_specialized.py.
//
// The kernel sources travel inside libnsgym_hip.so (.incbin below), so the library stays a
// self-contained drop-in; libhiprtc is dlopen'ed on first use (the generic path never needs it).
#pragma once

#include <dlfcn.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__)
#define NSG_EMBED(sym, file)                                                                      \
  __asm__(".pushsection .rodata\n.global " #sym "\n.type " #sym ", @object\n" #sym ":\n.incbin \"" file \
          "\"\n.byte 0\n.size " #sym ", .-" #sym "\n.popsection\n")
NSG_EMBED(nsg_src_abi, "../../include/nsgym_hip.h");
NSG_EMBED(nsg_src_math, "nsg_math.hip.h");
NSG_EMBED(nsg_src_libm, "nsg_libm.hip.h");
NSG_EMBED(nsg_src_sincos_tab, "../../include/nsg_sincos_tab.inc");
NSG_EMBED(nsg_src_pow_tab, "../../include/nsg_pow_tab.inc");
NSG_EMBED(nsg_src_powf_tab, "../../include/nsg_powf_tab.inc");
NSG_EMBED(nsg_src_rng, "nsg_rng.hip.h");
NSG_EMBED(nsg_src_theta, "nsg_theta.hip.h");
NSG_EMBED(nsg_src_envs, "nsg_envs.hip.h");
NSG_EMBED(nsg_src_kernels, "nsg_kernels.hip.h");
NSG_EMBED(nsg_src_rollout, "nsg_rollout.hip.h");
#endif
extern "C" {
extern const char nsg_src_abi[], nsg_src_math[], nsg_src_libm[], nsg_src_sincos_tab[], nsg_src_pow_tab[], nsg_src_powf_tab[], nsg_src_rng[], nsg_src_theta[], nsg_src_envs[],
    nsg_src_kernels[], nsg_src_rollout[];
}

namespace nsg_spec {

// ---- hiprtc through dlopen ---------------------------------------------------------------------
typedef struct _hiprtcProgram* rtcProgram;
struct Rtc {
  void* lib = nullptr;
  int (*create)(rtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*compile)(rtcProgram, int, const char* const*) = nullptr;
  int (*log_size)(rtcProgram, size_t*) = nullptr;
  int (*log)(rtcProgram, char*) = nullptr;
  int (*code_size)(rtcProgram, size_t*) = nullptr;
  int (*code)(rtcProgram, char*) = nullptr;
  int (*destroy)(rtcProgram*) = nullptr;
};

inline const Rtc* rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    if (const char* off = getenv("NSG_NO_HIPRTC"))   // test hook: behave like a box without the runtime compiler
      if (off[0] == '1') return;
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
#define NSG_SYM(field, name) r.field = (decltype(r.field))dlsym(r.lib, name)
    NSG_SYM(create, "hiprtcCreateProgram");
    NSG_SYM(compile, "hiprtcCompileProgram");
    NSG_SYM(log_size, "hiprtcGetProgramLogSize");
    NSG_SYM(log, "hiprtcGetProgramLog");
    NSG_SYM(code_size, "hiprtcGetCodeSize");
    NSG_SYM(code, "hiprtcGetCode");
    NSG_SYM(destroy, "hiprtcDestroyProgram");
#undef NSG_SYM
    if (!r.create || !r.compile || !r.log_size || !r.log || !r.code_size || !r.code || !r.destroy) {
      dlclose(r.lib);
      r.lib = nullptr;
    }
  });
  return r.lib ? &r : nullptr;
}

// ---- the specialised translation unit ----------------------------------------------------------
// The config is emitted as its raw 64-bit words (independent of the struct's field list) and viewed
// as nsg_config; after inlining every access is a load from a constant at a constant offset.
constexpr const char* kGeneratorRev = "spec_source r3.4";   // part of the cache key (nsgym_hip.hip: spec_source_hash)
// Does the config look anything up in the constant-table blob (schedule bit tables, value lists, grid maps)?
inline bool cfg_uses_table_blob(const nsg_config& cfg) {
  if (cfg.env_type == NSG_ENV_FROZENLAKE || cfg.env_type == NSG_ENV_CLIFFWALKING || cfg.env_type == NSG_ENV_BRIDGE) return true;
  for (int p = 0; p < cfg.n_params; p++) {
    const nsg_param_cfg& pc = cfg.params[p];
    if (pc.sched_kind == NSG_SCHED_TABLE || pc.val_tab_len > 0) return true;
  }
  return false;
}
inline std::string spec_source(const nsg_config& cfg, bool full, bool resets_in_lane = false, bool six_waves = true, bool stream_state = false) {
  static_assert(sizeof(nsg_config) % 8 == 0, "nsg_config is emitted as 64-bit words");
  std::string s;
  s.reserve(16384);
  s +=
      "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
      "typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t;\n"
      "typedef unsigned long size_t;\n"
      "#define NSG_SPEC_BUILD 1\n";
  if (resets_in_lane) s += "#define NSG_CARTPOLE_INLANE 1\n";   // batch-size policy of nsg_specialize (nsg_envs.hip.h)
  if (cfg.flags & NSG_F_LIBM_EXACT) s += "#define NSG_LIBM_EXACT 1\n";   // the integrators' sin / cos: libm's, bit for bit (nsg_libm.hip.h)
  if (stream_state) s += "#ifndef NSG_STREAM_STATE\n#define NSG_STREAM_STATE 1\n#endif\n";   // likewise (nsg_rng.hip.h: stg)
  // a classic-control config without a table blob (no schedule bit table, no value list): the step kernel reads the ziggurat tables
  // where they are instead of staging them per workgroup (nsg_kernels.hip.h: stage_tables<DIRECT>)
  if (!cfg_uses_table_blob(cfg)) s += "#ifndef NSG_TABLES_DIRECT\n#define NSG_TABLES_DIRECT 1\n#endif\n";
  // CartPole's step is asked to keep 6 wavefronts per SIMD (<= 80 VGPRs): the launch policy (step_grid_for: 6 workgroups per
  // CU, all resident from the start) is built on it, and at 81 VGPRs the same kernel is 8 % slower (C1 2^20 envs: 23.9 ->
  // 26.3 us).  The compiler meets the bound without spilling for the usual configs (C1 69, C2 75 VGPRs); a config for which
  // it cannot is rebuilt without the bound (spec_compile).
  if (six_waves && cfg.env_type == NSG_ENV_CARTPOLE) s += "#ifndef NSG_MIN_WAVES\n#define NSG_MIN_WAVES 6\n#endif\n";
  s +=
      "#include \"nsg_rollout.hip.h\"\n"
      "namespace nsg {\n"
      "__device__ const uint64_t kCfgWords[] = {\n";
  const uint64_t* w = reinterpret_cast<const uint64_t*>(&cfg);
  char buf[40];
  for (size_t k = 0; k < sizeof(nsg_config) / 8; k++) {
    snprintf(buf, sizeof(buf), "0x%016llxull,%s", (unsigned long long)w[k], (k % 4 == 3) ? "\n" : " ");
    s += buf;
  }
  s += "};\n}  // namespace nsg\n";
  snprintf(buf, sizeof(buf), "%d, %s", (int)cfg.env_type, full ? "true" : "false");
  const std::string targs = buf;
  s += "#define NSG_SPEC_CFG (*reinterpret_cast<const nsg_config*>(nsg::kCfgWords))\n"
       "extern \"C\" __global__ __launch_bounds__(NSG_BLOCK, NSG_MIN_WAVES) void nsg_spec_step(const nsg::Segment* __restrict__ seg,\n"
       "                                                                const void* __restrict__ actions, int reverse) {\n"
       "  nsg::step_body<" + targs + ">(NSG_SPEC_CFG, *seg, actions, (int)blockIdx.x, (int)gridDim.x, reverse);\n"
       "}\n"
       "extern \"C\" __global__ __launch_bounds__(NSG_BLOCK) void nsg_spec_rollout(const nsg::Segment* __restrict__ seg,\n"
       "                                                                   const void* __restrict__ actions, int k_steps,\n"
       "                                                                   nsg_rollout_out ro) {\n"
       "  nsg::rollout_body<" + targs + ">(NSG_SPEC_CFG, *seg, actions, k_steps, ro, (int)blockIdx.x, (int)gridDim.x);\n"
       "}\n";
  return s;
}

// The resident stepper (nsg_resident_start) of a specialised handle: its own small unit, compiled the first time a resident loop
// is started on such a handle (the step / rollout units of everybody else stay as they are).
inline std::string resident_source(const nsg_config& cfg, bool full) {
  std::string s = spec_source(cfg, full, false, false, false);
  const size_t cut = s.find("extern \"C\" __global__");
  s.resize(cut);
  char buf[40];
  snprintf(buf, sizeof(buf), "%d, %s", (int)cfg.env_type, full ? "true" : "false");
  s += "extern \"C\" __global__ __launch_bounds__(NSG_BLOCK) void nsg_spec_resident(const nsg::Segment* __restrict__ seg,\n"
       "                                                                    const void* __restrict__ actions, nsg::ResidentArgs ra) {\n"
       "  nsg::resident_body<" + std::string(buf) + ">(NSG_SPEC_CFG, *seg, actions, ra);\n"
       "}\n";
  return s;
}

// The fused policy rollout (nsg_rollout_policy) of a specialised handle: its own small unit like the resident stepper's - one per action
// source (`kind`, a compile-time constant of the unit) -, compiled the first time such a rollout is asked for (with the batch-size policy of the handle's main unit, so that its resets and stores behave
// like the handle's nsg_rollout).
inline std::string policy_source(const nsg_config& cfg, bool full, bool resets_in_lane, bool stream_state, int kind) {
  std::string s = spec_source(cfg, full, resets_in_lane, false, stream_state);
  const size_t cut = s.find("extern \"C\" __global__");
  s.resize(cut);
  char buf[48];
  snprintf(buf, sizeof(buf), "%d, %s, true, %d", (int)cfg.env_type, full ? "true" : "false", kind);
  s += "extern \"C\" __global__ __launch_bounds__(NSG_BLOCK) void nsg_spec_rollout_policy(const nsg::Segment* __restrict__ seg, int k_steps,\n"
       "                                                                          nsg_rollout_out ro, nsg::PolicyArgs pa) {\n"
       "  nsg::rollout_body<" + std::string(buf) + ">(NSG_SPEC_CFG, *seg, pa.pol.data, k_steps, ro, (int)blockIdx.x, (int)gridDim.x, &pa);\n"
       "}\n";
  return s;
}

inline void emit_cfg_words(std::string& s, const nsg_config& cfg, int index) {
  char buf[48];
  snprintf(buf, sizeof(buf), "__device__ const uint64_t kCfgWords%d[] = {\n", index);
  s += buf;
  const uint64_t* w = reinterpret_cast<const uint64_t*>(&cfg);
  for (size_t k = 0; k < sizeof(nsg_config) / 8; k++) {
    snprintf(buf, sizeof(buf), "0x%016llxull,%s", (unsigned long long)w[k], (k % 4 == 3) ? "\n" : " ");
    s += buf;
  }
  s += "};\n";
}

// Heterogeneous launch (nsg_step_group) specialised for the ordered tuple of its members' configs:
// one kernel, the segment a workgroup belongs to selects the member's folded step_body.
inline std::string group_source(const nsg_config* const* cfgs, const bool* full, int n, bool with_rollout = true) {
  std::string s;
  s.reserve(16384 * n);
  s +=
      "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
      "typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t;\n"
      "typedef unsigned long size_t;\n"
      "#define NSG_SPEC_BUILD 1\n";
  // classic-control members read their tables where they are when none of them uses a table blob (stage_tables<DIRECT>; grid
  // members keep staging their maps either way)
  bool direct = true;
  for (int k = 0; k < n; k++) {
    const int e = cfgs[k]->env_type;
    const bool grid = e == NSG_ENV_FROZENLAKE || e == NSG_ENV_CLIFFWALKING || e == NSG_ENV_BRIDGE;
    if (!grid && cfg_uses_table_blob(*cfgs[k])) direct = false;
  }
  if (direct) s += "#ifndef NSG_TABLES_DIRECT\n#define NSG_TABLES_DIRECT 1\n#endif\n";
  // libm's arithmetic is a property of the translation unit: a group is exact when every classic-control member is (the caller has
  // refused mixed lists: check_group_members)
  for (int k = 0; k < n; k++)
    if (cfgs[k]->flags & NSG_F_LIBM_EXACT) { s += "#define NSG_LIBM_EXACT 1\n"; break; }
  s +=
      "#include \"nsg_rollout.hip.h\"\n"
      "namespace nsg {\n";
  for (int k = 0; k < n; k++) emit_cfg_words(s, *cfgs[k], k);
  s += "}  // namespace nsg\n"
       "extern \"C\" __global__ __launch_bounds__(NSG_BLOCK, NSG_MIN_WAVES) void nsg_spec_group(const nsg::Segment* __restrict__ segs, int nseg,\n"
       "                                                                 nsg::ActionPtrs acts, int reverse) {\n"
       "  const int sidx = nsg::group_segment_of_block(segs, nseg);\n"
       "  const nsg::Segment& sg = segs[sidx];\n"
       "  const int rel = (int)blockIdx.x - sg.block_begin;\n"
       "  switch (sidx) {\n";
  char buf[256];
  for (int k = 0; k < n; k++) {
    snprintf(buf, sizeof(buf),
             "    case %d: nsg::step_body<%d, %s>(*reinterpret_cast<const nsg_config*>(nsg::kCfgWords%d), sg, acts.p[%d], rel, "
             "sg.block_count, reverse); break;\n",
             k, (int)cfgs[k]->env_type, full[k] ? "true" : "false", k, k);
    s += buf;
  }
  s += "    default: break;\n  }\n}\n";
  if (!with_rollout) return s;
  s += "extern \"C\" __global__ __launch_bounds__(NSG_BLOCK) void nsg_spec_group_rollout(const nsg::Segment* __restrict__ segs, int nseg,\n"
       "                                                                 nsg::ActionPtrs acts, int k_steps, nsg::RolloutOuts outs) {\n"
       "  const int sidx = nsg::group_segment_of_block(segs, nseg);\n"
       "  const nsg::Segment& sg = segs[sidx];\n"
       "  const int rel = (int)blockIdx.x - sg.block_begin;\n"
       "  switch (sidx) {\n";
  for (int k = 0; k < n; k++) {
    snprintf(buf, sizeof(buf),
             "    case %d: nsg::rollout_body<%d, %s>(*reinterpret_cast<const nsg_config*>(nsg::kCfgWords%d), sg, acts.p[%d], k_steps, "
             "outs.o[%d], rel, sg.block_count); break;\n",
             k, (int)cfgs[k]->env_type, full[k] ? "true" : "false", k, k, k);
    s += buf;
  }
  s += "    default: break;\n  }\n}\n";
  return s;
}

inline std::vector<char> compile_source(const std::string& src, const char* arch, std::string& err);

// Largest .vgpr_spill_count / .private_segment_fixed_size over ALL kernels of a code object (-1: metadata not found).  A unit
// is judged as a whole: its step, rollout and group kernels are all product paths.
inline int max_metadata_value(const std::vector<char>& code, const char* key17or27) {
  const std::string blob(code.begin(), code.end());
  const size_t klen = strlen(key17or27);
  std::string k;
  if (klen < 32) k += (char)(0xa0 + klen);              // fixstr
  else { k += (char)0xd9; k += (char)klen; }            // str8
  k += key17or27;
  int worst = -1;
  for (size_t b = blob.find(k); b != std::string::npos; b = blob.find(k, b + 1)) {
    if (b + k.size() >= blob.size()) break;
    const unsigned char* v = (const unsigned char*)blob.data() + b + k.size();
    const size_t left = blob.size() - (b + k.size());
    int x = -1;
    if (v[0] < 0x80) x = v[0];
    else if (v[0] == 0xcc && left >= 2) x = v[1];
    else if (v[0] == 0xcd && left >= 3) x = (v[1] << 8) | v[2];
    else if (v[0] == 0xce && left >= 5) x = (int)(((unsigned)v[1] << 24) | (v[2] << 16) | (v[3] << 8) | v[4]);
    if (x > worst) worst = x;
  }
  return worst;
}
// true when some kernel of the unit spills vector registers or owns scratch memory (or the metadata cannot be read)
inline bool unit_uses_scratch(const std::vector<char>& code) {
  return max_metadata_value(code, ".vgpr_spill_count") != 0 || max_metadata_value(code, ".private_segment_fixed_size") != 0;
}
// NSG_SPEC_ALLOW_SPILL=1: DIAGNOSTIC ONLY (tools/spill_probe.py) - keep a unit that spills instead of rebuilding / refusing it
inline bool allow_spill() {
  const char* e = getenv("NSG_SPEC_ALLOW_SPILL");
  return e && e[0] == '1';
}

// Compile the specialised unit for `arch` (e.g. "gfx950"); no GPU needed.  Returns "" and fills `err` on failure.
// A unit in which ANY kernel - step or rollout - spills vector registers or uses scratch memory is never shipped: spilling costs
// more than the occupancy the register bound was asked for (a two-update-fn CartPole config with its own stream: 54-100 spilled
// VGPRs, 164-228 B of scratch per lane), and one such build returned wrong results on MI355X (round 2, random-configuration
// case 61; profiles/r03_case61_spill_evidence.md).  Such a config is compiled again without the bound; if it still spills
// (only reachable through NSG_SPEC_FLAGS forcing a register bound) the unit is refused and the generic kernels stay in force.
inline std::vector<char> spec_compile(const nsg_config& cfg, bool full, const char* arch, std::string& err, bool resets_in_lane = false,
                                      bool stream_state = false) {
  std::vector<char> code = compile_source(spec_source(cfg, full, resets_in_lane, true, stream_state), arch, err);
  if (!code.empty() && unit_uses_scratch(code) && !allow_spill())
    code = compile_source(spec_source(cfg, full, resets_in_lane, false, stream_state), arch, err);
  if (!code.empty() && unit_uses_scratch(code) && !allow_spill()) {
    err = "a kernel of the specialised unit spills vector registers (scratch memory) under the given NSG_SPEC_FLAGS; such builds "
          "are not used (see spec_compile)";
    code.clear();
  }
  return code;
}
// The heterogeneous launch's unit is held to the same rule: a group kernel that spills is refused (the generic group kernel
// stays in force, nsg_step_group).
inline std::vector<char> resident_compile(const nsg_config& cfg, bool full, const char* arch, std::string& err) {
  std::vector<char> code = compile_source(resident_source(cfg, full), arch, err);
  if (!code.empty() && unit_uses_scratch(code) && !allow_spill()) {
    err = "the specialised resident kernel spills vector registers (scratch memory); such builds are not used (see spec_compile)";
    code.clear();
  }
  return code;
}
inline std::vector<char> policy_compile(const nsg_config& cfg, bool full, int kind, const char* arch, std::string& err, bool resets_in_lane = false,
                                        bool stream_state = false) {
  std::vector<char> code = compile_source(policy_source(cfg, full, resets_in_lane, stream_state, kind), arch, err);
  if (!code.empty() && unit_uses_scratch(code) && !allow_spill()) {
    err = "the specialised policy-rollout kernel spills vector registers (scratch memory); such builds are not used (see spec_compile)";
    code.clear();
  }
  return code;
}
// The two kernels of the unit are judged separately: when the unit with both spills, the single-step kernel is built on its own -
// if IT is clean the unit ships without nsg_spec_group_rollout (fused group rollouts of this member list run the generic kernel),
// so a spilling rollout never costs nsg_step_group its specialised kernel.
inline std::vector<char> group_compile(const nsg_config* const* cfgs, const bool* full, int n, const char* arch, std::string& err) {
  std::vector<char> code = compile_source(group_source(cfgs, full, n), arch, err);
  if (!code.empty() && unit_uses_scratch(code) && !allow_spill())
    code = compile_source(group_source(cfgs, full, n, false), arch, err);
  if (!code.empty() && unit_uses_scratch(code) && !allow_spill()) {
    err = "the specialised group kernel spills vector registers (scratch memory); such builds are not used (see spec_compile)";
    code.clear();
  }
  return code;
}

inline std::vector<char> compile_source(const std::string& src, const char* arch, std::string& err) {
  std::vector<char> code;
  const Rtc* r = rtc();
  if (!r) {
    err = "libhiprtc.so could not be loaded (config specialisation needs the ROCm runtime compiler)";
    return code;
  }
  const char* headers[] = {nsg_src_abi, nsg_src_math, nsg_src_libm, nsg_src_sincos_tab, nsg_src_pow_tab, nsg_src_powf_tab, nsg_src_rng, nsg_src_theta,
                           nsg_src_envs, nsg_src_kernels, nsg_src_rollout};
  const char* names[] = {"nsgym_hip.h",   "nsg_math.hip.h",  "nsg_libm.hip.h",    "nsg_sincos_tab.inc", "nsg_pow_tab.inc", "nsg_powf_tab.inc", "nsg_rng.hip.h",
                         "nsg_theta.hip.h", "nsg_envs.hip.h", "nsg_kernels.hip.h", "nsg_rollout.hip.h"};
  rtcProgram prog = nullptr;
  if (r->create(&prog, src.c_str(), "nsg_spec.hip", 11, headers, names) != 0) {
    err = "hiprtcCreateProgram failed";
    return code;
  }
  if (const char* dump = getenv("NSG_SPEC_DUMP")) {  // debugging aid: the generated translation unit
    if (FILE* f = fopen(dump, "w")) {
      fputs(src.c_str(), f);
      fclose(f);
    }
  }
  const std::string archopt = std::string("--offload-arch=") + arch;
  // the same flags as the library build (csrc/Makefile): contraction off keeps the float64 rounding sequence
  // the unit's workgroup size is the library's (the host launches both with kBlock threads)
  static const std::string blockopt = "-DNSG_BLOCK=" + std::to_string(NSG_BLOCK);
  // kernel arguments preloaded into SGPRs at wavefront launch (gfx940+; the compiler keeps a compatible entry for firmware without
  // it): one dependent scalar-load round trip less ahead of the first row load - C2 at 65 536 envs 6.91 -> 6.52 us, C1 5.01 -> 4.84
  std::vector<const char*> opts = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function", blockopt.c_str(),
                                   "-mllvm", "-amdgpu-kernarg-preload-count=4"};
  // tuning knob (tools/ab.py): extra -D / -m options for the specialised unit, space-separated
  std::vector<std::string> extra;
  if (const char* e = getenv("NSG_SPEC_FLAGS")) {
    std::string cur;
    for (const char* c = e;; c++) {
      if (*c == ' ' || *c == '\0') {
        if (!cur.empty()) extra.push_back(cur);
        cur.clear();
        if (!*c) break;
      } else {
        cur += *c;
      }
    }
  }
  for (const std::string& x : extra) {
    if (x.compare(0, 8, "-DNSG_X_") == 0) {   // the timing ablations of earlier rounds (wrong results by design) are gone from the kernels
      err = "NSG_SPEC_FLAGS: " + x + " names a timing ablation; those are not part of the product kernels";
      r->destroy(&prog);
      return code;
    }
    opts.push_back(x.c_str());
  }
  const int rc = r->compile(prog, (int)opts.size(), opts.data());
  if (rc != 0) {
    size_t n = 0;
    r->log_size(prog, &n);
    std::string log(n + 1, '\0');
    if (n) r->log(prog, &log[0]);
    err = "hiprtc compilation failed:\n" + log.substr(0, 1500);
    r->destroy(&prog);
    return code;
  }
  size_t n = 0;
  r->code_size(prog, &n);
  code.resize(n);
  if (n) r->code(prog, code.data());
  r->destroy(&prog);
  if (code.empty()) err = "hiprtc produced an empty code object";
  return code;
}

inline uint64_t fnv1a(const void* p, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t k = 0; k < n; k++) h = (h ^ b[k]) * 0x100000001b3ull;
  return h;
}

// One loaded code object per (device, config): shared by every handle with the same configuration.
struct Module {
  hipModule_t mod = nullptr;
  hipFunction_t step = nullptr, rollout = nullptr;  // single-config unit
  hipFunction_t group = nullptr, group_rollout = nullptr;   // heterogeneous-launch unit (single step, fused rollout)
  hipFunction_t resident = nullptr;                         // resident-stepper unit
  hipFunction_t rollout_policy = nullptr;                   // fused policy-rollout unit
  uint64_t h0 = 0;                                    // config key (group keys are built from their members')
  int step_waves = 0;                                 // wavefronts per SIMD the step kernel's registers allow (0 = unknown)
  int origin = 0;                                     // NSG_SPEC_ORIGIN_*: where this code object came from
};

struct Key {
  int device;
  uint64_t h0, h1;
  bool operator<(const Key& o) const {
    return device != o.device ? device < o.device : h0 != o.h0 ? h0 < o.h0 : h1 < o.h1;
  }
};

inline std::map<Key, Module>& cache() {
  static std::map<Key, Module> m;
  return m;
}
inline std::mutex& cache_mutex() {
  static std::mutex m;
  return m;
}
// keys whose compilation failed in this process (refused unit, compiler error): asked again they fail at once instead of
// compiling for seconds every time - a re-plan of a group holds the plan mutex while it asks
inline std::map<Key, std::string>& failed() {
  static std::map<Key, std::string> m;
  return m;
}

}  // namespace nsg_spec
