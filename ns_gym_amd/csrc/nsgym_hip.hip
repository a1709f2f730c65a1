// nsgym_hip.hip — C-ABI host side of libnsgym_hip.so (see include/nsgym_hip.h).
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <sys/stat.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <new>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>

#include "nsg_zig_tables.inc"
#include "nsgym_hip.h"
#include "nsg_kernels.hip.h"
#include "nsg_rollout.hip.h"
#include "nsg_specialize.host.h"

using namespace nsg;


namespace {

thread_local char g_err[512] = "";
std::atomic<uint64_t> g_next_handle_id{1};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return fail(NSG_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

const int kPhysDim[NSG_ENV_COUNT] = {4, 2, 4, 2, 2, 0, 0, 0};
const int kObsDim[NSG_ENV_COUNT] = {4, 3, 6, 2, 2, 1, 1, 1};
const int kNTheta[NSG_ENV_COUNT] = {6, 4, 8, 2, 1, 1, 1, 3};
const int kNActions[NSG_ENV_COUNT] = {2, 0, 3, 3, 0, 4, 4, 4};
const int kNDist[NSG_ENV_COUNT] = {0, 0, 0, 0, 0, 3, 4, 3};

bool upd_is_normal(int k) {
  return k == NSG_UPD_RANDOMWALK || k == NSG_UPD_RW_DRIFT || k == NSG_UPD_RW_DRIFT_TREND || k == NSG_UPD_OU ||
         k == NSG_UPD_BOUNDED_RW;
}

int validate(const nsg_config* cfg, size_t table_bytes) {
  if (!cfg) return fail(NSG_EINVAL, "cfg is NULL");
  if (cfg->abi_version != NSG_ABI_VERSION) return fail(NSG_EINVAL, "abi_version %d != %d", cfg->abi_version, NSG_ABI_VERSION);
  if (cfg->env_type < 0 || cfg->env_type >= NSG_ENV_COUNT) return fail(NSG_EINVAL, "bad env_type %d", cfg->env_type);
  if (cfg->n_params < 0 || cfg->n_params > NSG_MAX_PARAMS) return fail(NSG_EINVAL, "bad n_params %d", cfg->n_params);
  if (table_bytes > (size_t)kMaxTableBytes) return fail(NSG_EINVAL, "constant tables are %zu bytes, limit %d", table_bytes, kMaxTableBytes);
  if (cfg->flags & ~NSG_F_KNOWN) return fail(NSG_EINVAL, "unknown flag bits 0x%x (this library knows 0x%x)", cfg->flags & ~NSG_F_KNOWN, NSG_F_KNOWN);
  if ((cfg->flags & NSG_F_LIBM_EXACT) && is_grid_env(cfg->env_type))
    return fail(NSG_EINVAL, "NSG_F_LIBM_EXACT is for the classic-control envs (the grid envs' path is integer arithmetic: nothing to choose)");
  if ((cfg->flags & NSG_F_NO_AUTORESET) && (cfg->flags & NSG_F_TRACK_RETURNS))
    return fail(NSG_EINVAL, "NSG_F_TRACK_RETURNS needs the autoreset (an episode's return is closed by the step that resets it); not combinable with NSG_F_NO_AUTORESET");
  const bool fl = is_grid_env(cfg->env_type);
  if (fl) {
    if (cfg->env_type != NSG_ENV_BRIDGE && cfg->n_params != 1) return fail(NSG_EINVAL, "FrozenLake / CliffWalking take exactly one tunable parameter (P)");
    if (cfg->env_type == NSG_ENV_BRIDGE && (cfg->n_params < 1 || cfg->n_params > 2)) return fail(NSG_EINVAL, "Bridge takes P, or P_left and/or P_right");
    if (cfg->nrow <= 0 || cfg->ncol <= 0) return fail(NSG_EINVAL, "bad grid map %dx%d", cfg->nrow, cfg->ncol);
    if ((size_t)cfg->desc_tab_off + (size_t)cfg->nrow * cfg->ncol > table_bytes) return fail(NSG_EINVAL, "desc table out of range");
  }
  unsigned seen = 0;
  for (int p = 0; p < cfg->n_params; p++) {
    const nsg_param_cfg& pc = cfg->params[p];
    if (pc.theta_slot < 0 || pc.theta_slot >= kNTheta[cfg->env_type]) return fail(NSG_EINVAL, "param %d: bad theta_slot %d", p, pc.theta_slot);
    if (!fl && (seen & (1u << pc.theta_slot))) return fail(NSG_EINVAL, "param %d: theta_slot %d configured twice", p, pc.theta_slot);
    seen |= 1u << pc.theta_slot;
    const bool dist = pc.upd_kind >= NSG_UPD_D_INCREMENT;
    if (dist != fl) return fail(NSG_EINVAL, "param %d: update kind %d does not fit env type %d", p, pc.upd_kind, cfg->env_type);
    if (dist ? pc.upd_kind > NSG_UPD_D_LCBOUNDED : (pc.upd_kind < 0 || pc.upd_kind > NSG_UPD_BOUNDED_RW))
      return fail(NSG_EINVAL, "param %d: unknown update kind %d", p, pc.upd_kind);
    switch (pc.sched_kind) {
      case NSG_SCHED_CONTINUOUS: break;
      case NSG_SCHED_PERIODIC:
        if (pc.sched_i0 <= 0 || pc.sched_i0 > 0x7fffffff) return fail(NSG_EINVAL, "param %d: period must be in [1, 2^31)", p);
        break;
      case NSG_SCHED_BURST:
        if (pc.sched_i0 < 0 || pc.sched_i1 < 0 || pc.sched_i0 + pc.sched_i1 <= 0 || pc.sched_i0 + pc.sched_i1 > 0x7fffffff)
          return fail(NSG_EINVAL, "param %d: bad burst durations", p);
        break;
      case NSG_SCHED_TABLE:
        if (pc.sched_tab_len < 0 || pc.sched_tab_off < 0 ||
            (size_t)pc.sched_tab_off * 4 + ((size_t)pc.sched_tab_len + 31) / 32 * 4 > table_bytes)
          return fail(NSG_EINVAL, "param %d: schedule bit table out of range", p);
        break;
      case NSG_SCHED_RANDOM:
      case NSG_SCHED_DECAYING:
        if (!(pc.sched_p0 == pc.sched_p0)) return fail(NSG_EINVAL, "param %d: scheduler probability is NaN", p);
        break;
      case NSG_SCHED_MEMORYLESS:
        if (!(pc.sched_p0 > 0.0 && pc.sched_p0 <= 1.0)) return fail(NSG_EINVAL, "param %d: Memoryless p must be in (0, 1]", p);
        break;
      default: return fail(NSG_EINVAL, "param %d: scheduler kind %d is not supported by the kernels", p, pc.sched_kind);
    }
    const int k = pc.upd_kind;
    const bool needs_tab = k == NSG_UPD_POLY || k == NSG_UPD_STEPWISE || k == NSG_UPD_CYCLIC || k == NSG_UPD_D_STEPWISE || k == NSG_UPD_D_CYCLIC;
    if (needs_tab) {
      const size_t per = dist ? (size_t)kNDist[cfg->env_type] : 1;
      if (pc.val_tab_len < 0 || pc.val_tab_off < 0 || ((size_t)pc.val_tab_off + (size_t)pc.val_tab_len * per) * 8 > table_bytes)
        return fail(NSG_EINVAL, "param %d: value table out of range", p);
      if ((k == NSG_UPD_CYCLIC || k == NSG_UPD_D_CYCLIC) && pc.val_tab_len == 0) return fail(NSG_EINVAL, "param %d: empty cyclic list", p);
    }
    if (k != NSG_UPD_D_LCBOUNDED && pc.uses_rng != ((upd_is_normal(k) || k == NSG_UPD_D_RANDOMCAT) ? 1 : 0)) return fail(NSG_EINVAL, "param %d: uses_rng does not match update kind %d", p, k);
    // shared objects: the slot is the FIRST entry using the same object, so it cannot point forward and must agree in kind
    if (pc.fn_slot < 0 || pc.fn_slot > p || cfg->params[pc.fn_slot].upd_kind != k || cfg->params[pc.fn_slot].uses_rng != pc.uses_rng ||
        cfg->params[pc.fn_slot].fn_slot != pc.fn_slot)
      return fail(NSG_EINVAL, "param %d: bad fn_slot %d", p, pc.fn_slot);
    if (pc.sched_slot < 0 || pc.sched_slot > p || cfg->params[pc.sched_slot].sched_kind != pc.sched_kind ||
        cfg->params[pc.sched_slot].sched_slot != pc.sched_slot)
      return fail(NSG_EINVAL, "param %d: bad sched_slot %d", p, pc.sched_slot);
  }
  return NSG_OK;
}

}  // namespace

struct nsg_handle {
  Segment host;        // host copy of the device segment
  Segment* dev;        // device copy (read by the kernels through scalar loads)
  uint8_t* d_tables;
  uint64_t* d_zig;
  int64_t n;
  bool bound;
  int device;
  const nsg_spec::Module* spec;  // config-specialised step / rollout kernels (nsg_specialize), or NULL
  const nsg_spec::Module* spec_resident = nullptr;   // ... and its resident stepper (built on the first nsg_resident_start)
  const nsg_spec::Module* spec_policy_unit[4] = {nullptr, nullptr, nullptr, nullptr};   // ... and its fused policy rollouts, one unit per action
  bool spec_policy_tried[4] = {false, false, false, false};                              // source (NSG_POL_*), built on the first nsg_rollout_policy of that kind
  unsigned launches = 0;
  // What nsg_step_group remembers about a member list is keyed on these two: `id` is unique per nsg_create for the life of the
  // process (a new handle at a recycled address is a different member), `generation` counts the launch-relevant changes of
  // THIS handle (nsg_bind, nsg_specialize) - other handles coming and going (planning copies) leave a group's plan alone.
  uint64_t id = 0;
  std::atomic<uint64_t> generation{0};   // written by nsg_bind / nsg_specialize of this handle, read by group planning under its mutex
};

namespace {

#define NSG_STR2(x) #x
#define NSG_STR(x) NSG_STR2(x)
#define HIP_VERSION_STR NSG_STR(HIP_VERSION_MAJOR) "." NSG_STR(HIP_VERSION_MINOR) "." NSG_STR(HIP_VERSION_PATCH)

uint64_t spec_source_hash() {
  uint64_t h1 = 0x9e3779b97f4a7c15ull;
  for (const char* src : {nsg_src_abi, nsg_src_math, nsg_src_libm, nsg_src_sincos_tab, nsg_src_pow_tab, nsg_src_powf_tab, nsg_src_rng, nsg_src_theta, nsg_src_envs, nsg_src_kernels, nsg_src_rollout})
    h1 = nsg_spec::fnv1a(src, strlen(src), h1);
  if (const char* e = getenv("NSG_SPEC_FLAGS")) h1 = nsg_spec::fnv1a(e, strlen(e), h1);  // extra compile options are part of the key
  if (nsg_spec::allow_spill()) h1 = nsg_spec::fnv1a("allow-spill", 11, h1);              // a diagnostic build never shares a cache entry
  // so are the fixed options of nsg_spec::compile_source (keep this literal in step with them) and the toolchain the
  // library was built with: code objects persist on disk between processes (spec_cache_dir)
  static const char kFixed[] = "-O3 -std=c++17 -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=4 block " NSG_STR(NSG_BLOCK) " hip " HIP_VERSION_STR;
  h1 = nsg_spec::fnv1a(kFixed, sizeof(kFixed), h1);
  // and the text spec_source() wraps around the headers (launch bounds, per-env-type defines): bump when it changes
  h1 = nsg_spec::fnv1a(nsg_spec::kGeneratorRev, strlen(nsg_spec::kGeneratorRev), h1);
  return h1;
}

// Where compiled units persist between processes: NSG_SPEC_CACHE=<dir>, or "off" / "0" / "" for none; unset = the user's
// cache directory ($XDG_CACHE_HOME or $HOME/.cache)/ns_gym_amd.  Created on demand; any failure just means no disk cache.
std::string spec_cache_dir() {
  std::string dir;
  if (const char* e = getenv("NSG_SPEC_CACHE")) {
    if (!*e || !strcmp(e, "off") || !strcmp(e, "0")) return "";
    dir = e;
  } else if (const char* x = getenv("XDG_CACHE_HOME")) {
    if (*x) dir = std::string(x) + "/ns_gym_amd";
  }
  if (dir.empty()) {
    const char* home = getenv("HOME");
    if (!home || !*home) return "";
    dir = std::string(home) + "/.cache";
    (void)mkdir(dir.c_str(), 0755);
    dir += "/ns_gym_amd";
  }
  (void)mkdir(dir.c_str(), 0700);   // code objects are loaded onto the GPU: nobody else gets to put files here
  // refuse a directory that somebody else owns or can write to (a shared NSG_SPEC_CACHE must not let another user plant device code)
  struct stat st;
  if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return "";
  return dir;
}

// A cached object is used only if it is a regular file of this user that no one else may write.
bool spec_cache_file_ok(const char* path) {
  struct stat st;
  return lstat(path, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH));
}

// Keep the disk cache bounded: every distinct configuration (constructor seeds and update constants included - they are folded
// into the unit) adds a ~60-KB file.  Beyond kSpecCacheMaxFiles the least recently used ones go (a cache hit refreshes a
// file's timestamp).
constexpr size_t kSpecCacheMaxFiles = 256;
void spec_cache_evict(const std::string& dir) {
  std::vector<std::pair<time_t, std::string>> files;
  if (DIR* d = opendir(dir.c_str())) {
    while (struct dirent* e = readdir(d)) {
      const std::string name = e->d_name;
      if (name.size() < 10 || name.compare(0, 4, "nsg_") != 0 || name.compare(name.size() - 6, 6, ".hsaco") != 0) continue;
      struct stat st;
      const std::string path = dir + "/" + name;
      if (lstat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode)) files.emplace_back(st.st_mtime, path);
    }
    closedir(d);
  }
  if (files.size() <= kSpecCacheMaxFiles) return;
  std::sort(files.begin(), files.end());
  for (size_t k = 0; k + kSpecCacheMaxFiles < files.size(); k++) (void)remove(files[k].second.c_str());
}

// Units shipped WITH the library, built and inspected when the library was (nsg_spec_prebuild; csrc/prebuilt_resource_usage.txt):
// <directory of libnsgym_hip.so>/prebuilt, or NSG_PREBUILT_DIR.  Same trust as the library next to them: no ownership test.
std::string prebuilt_dir() {
  if (const char* e = getenv("NSG_PREBUILT_DIR")) return (*e && strcmp(e, "off") != 0) ? std::string(e) : std::string();
  Dl_info info;
  if (!dladdr((const void*)&nsg_abi_version, &info) || !info.dli_fname) return "";
  std::string path = info.dli_fname;
  const size_t slash = path.rfind('/');
  return (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/prebuilt";
}

std::string spec_file_name(uint64_t h0, uint64_t h1) {
  char name[64];
  snprintf(name, sizeof(name), "/nsg_%016llx_%016llx.hsaco", (unsigned long long)h0, (unsigned long long)h1);
  return name;
}

bool read_file(const std::string& path, std::vector<char>& code) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  code.clear();
  if (n > 0) {
    code.resize((size_t)n);
    if (fread(code.data(), 1, (size_t)n, f) != (size_t)n) code.clear();
  }
  fclose(f);
  return !code.empty();
}

// Look a code object up in the process cache, then among the prebuilt units, then in the disk cache (spec_cache_dir), else
// compile it; load it.
enum { kUnitSingle = 0, kUnitGroup = 1, kUnitResident = 2, kUnitPolicy = 3 };
template <typename Compile>
int get_spec_module(int device, uint64_t h0, int unit_kind, Compile&& compile, const nsg_spec::Module** out) {
  const bool group = unit_kind == kUnitGroup;
  const uint64_t h1 = spec_source_hash();
  const nsg_spec::Key key{device, h0, h1};
  std::lock_guard<std::mutex> lock(nsg_spec::cache_mutex());
  auto& cache = nsg_spec::cache();
  auto it = cache.find(key);
  if (it == cache.end()) {
    auto bad = nsg_spec::failed().find(key);
    if (bad != nsg_spec::failed().end()) return fail(NSG_EUNSUPPORTED, "%s", bad->second.c_str());
    std::vector<char> code;
    std::string path;
    int origin = NSG_SPEC_ORIGIN_HIPRTC;
    const std::string pre = prebuilt_dir();
    if (!pre.empty() && read_file(pre + spec_file_name(h0, h1), code)) {
      nsg_spec::Module pm;
      pm.h0 = h0;
      if (hipModuleLoadData(&pm.mod, code.data()) == hipSuccess) {
        (void)hipModuleUnload(pm.mod);
        origin = NSG_SPEC_ORIGIN_PREBUILT;
      } else {   // built for another target: fall through to the cache / the compiler
        (void)hipGetLastError();
        code.clear();
      }
    }
    const std::string dir = origin == NSG_SPEC_ORIGIN_PREBUILT ? std::string() : spec_cache_dir();
    if (!dir.empty()) {
      path = dir + spec_file_name(h0, h1);
      if (FILE* f = spec_cache_file_ok(path.c_str()) ? fopen(path.c_str(), "rb") : nullptr) {
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (n > 0) {
          code.resize((size_t)n);
          if (fread(code.data(), 1, (size_t)n, f) != (size_t)n) code.clear();
        }
        fclose(f);
        if (!code.empty()) (void)utimes(path.c_str(), nullptr);   // least-recently-used eviction goes by this
      }
    }
    auto store = [&]() {  // through a temporary file + rename: concurrent processes (one per GPU) never see a partial object
      if (path.empty()) return;
      const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
      if (FILE* f = fopen(tmp.c_str(), "wb")) {
        const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
        fclose(f);
        (void)chmod(tmp.c_str(), 0600);
        if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
        else spec_cache_evict(dir);
      }
    };
    bool compiled_now = false;
    if (code.empty()) {
      std::string err;
      code = compile(err);
      if (code.empty()) {
        nsg_spec::failed()[key] = err;
        return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
      }
      compiled_now = true;
      store();
    } else if (origin != NSG_SPEC_ORIGIN_PREBUILT) {
      origin = NSG_SPEC_ORIGIN_CACHE;
    }
    nsg_spec::Module m;
    m.h0 = h0;
    const bool from_disk = !code.empty() && !path.empty() && !compiled_now;
    hipError_t le = hipModuleLoadData(&m.mod, code.data());
    if (le != hipSuccess && from_disk) {  // an unusable cached object (other GPU generation, damaged file): rebuild it
      (void)hipGetLastError();
      remove(path.c_str());
      std::string err;
      code = compile(err);
      if (code.empty()) {
        nsg_spec::failed()[key] = err;
        return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
      }
      store();
      origin = NSG_SPEC_ORIGIN_HIPRTC;
      le = hipModuleLoadData(&m.mod, code.data());
    }
    if (le != hipSuccess) return fail(NSG_EHIP, "hipModuleLoadData: %s", hipGetErrorString(le));
    m.origin = origin;
    if (unit_kind == kUnitResident) {
      HIP_TRY(hipModuleGetFunction(&m.resident, m.mod, "nsg_spec_resident"));
    } else if (unit_kind == kUnitPolicy) {
      HIP_TRY(hipModuleGetFunction(&m.rollout_policy, m.mod, "nsg_spec_rollout_policy"));
    } else if (group) {
      HIP_TRY(hipModuleGetFunction(&m.group, m.mod, "nsg_spec_group"));
      // a unit whose fused rollout would have spilled ships the single-step kernel alone (group_compile): rollouts of this member
      // list then run the generic kernel
      if (hipModuleGetFunction(&m.group_rollout, m.mod, "nsg_spec_group_rollout") != hipSuccess) {
        (void)hipGetLastError();
        m.group_rollout = nullptr;
      }
    } else {
      HIP_TRY(hipModuleGetFunction(&m.step, m.mod, "nsg_spec_step"));
      HIP_TRY(hipModuleGetFunction(&m.rollout, m.mod, "nsg_spec_rollout"));
      int regs = 0;   // the launch policy (step_grid_for) needs to know how many workgroups of this kernel a CU holds
      if (hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, m.step) == hipSuccess && regs > 0) {
        const int w = 512 / ((regs + 7) & ~7);
        m.step_waves = w > 8 ? 8 : w;
      }
    }
    it = cache.emplace(key, m).first;
  }
  *out = &it->second;
  return NSG_OK;
}

// What the unit of a (config, batch size) pair depends on besides the config itself, and its key.  Shared by nsg_specialize (the
// handle's own config and size, the device's gcnArchName) and nsg_spec_prebuild (the same for a device that is not there).
struct SpecPolicy {
  bool full;          // full theta engine (transcendentals / ziggurat) or the plain-arithmetic one
  bool inlane;        // CartPole batches of 2^16-2^17 envs (one wavefront per SIMD: the launch is bound by its serial chain) reset in-lane
                      // like the rare-reset env types - no hand-over, no barriers: C1 5.84 -> 5.60 us and C2 (BASELINE's own 65 536 envs)
                      // 7.80 -> 7.29 us at 2^16, +-0 at 2^17; smaller and larger batches lose (2^14: 5.3 -> 6.0, 2^18: 9.4 -> 10.1, 2^20: 23.7 -> 26.1)
  bool stream_state;  // classic-control batches of 2^24 envs and more (2.5 GB of rows: nothing a launch writes is still in the 256-MiB
                      // Infinity Cache when the next launch reads it) store their persistent rows non-temporally as well: C1 359.9 -> 348.6 us
                      // at 2^24 envs, C2 634.9 -> 610.3, Pendulum 298.5 -> 290.7 (+-0 at 2^23 and below, and for the grid envs, whose rows
                      // already leave through agent-scope stores)
};
bool cfg_simple_theta(const nsg_config& cfg) {
  for (int p = 0; p < cfg.n_params; p++)
    if (!upd_kind_is_simple(cfg.params[p].upd_kind) || sched_is_stochastic(cfg.params[p].sched_kind)) return false;
  return true;
}
SpecPolicy spec_policy(const nsg_config& cfg, int64_t n) {
  SpecPolicy p;
  p.full = !cfg_simple_theta(cfg);
  p.inlane = cfg.env_type == NSG_ENV_CARTPOLE && n >= 49152 && n <= 163840;
  p.stream_state = !is_grid_env(cfg.env_type) && n >= (1 << 24);
  return p;
}
// key: config bytes + engine variant + batch-size policy + target (+ the kernel sources this library was built from: spec_source_hash)
uint64_t spec_key(const nsg_config& cfg, const SpecPolicy& p, const char* arch_name) {
  uint64_t h0 = nsg_spec::fnv1a(&cfg, sizeof(nsg_config));
  h0 = nsg_spec::fnv1a(&p.full, sizeof(p.full), h0);
  h0 = nsg_spec::fnv1a(&p.inlane, sizeof(p.inlane), h0);
  h0 = nsg_spec::fnv1a(&p.stream_state, sizeof(p.stream_state), h0);
  h0 = nsg_spec::fnv1a(arch_name, strlen(arch_name), h0);
  if (const char* e = getenv("NSG_SPEC_FLAGS")) h0 = nsg_spec::fnv1a(e, strlen(e), h0);
  return h0;
}
uint64_t group_key(const uint64_t* member_keys, int n) {
  uint64_t h0 = 0x67726f7570ull;  // "group"
  for (int k = 0; k < n; k++) h0 = nsg_spec::fnv1a(&member_keys[k], sizeof(uint64_t), h0);
  return h0;
}
int write_unit(const char* dir, uint64_t h0, const std::vector<char>& code) {
  const std::string path = std::string(dir) + spec_file_name(h0, spec_source_hash());
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return fail(NSG_EINVAL, "cannot write %s", tmp.c_str());
  const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
  fclose(f);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
    remove(tmp.c_str());
    return fail(NSG_EINVAL, "cannot write %s", path.c_str());
  }
  return NSG_OK;
}
}  // namespace

extern "C" {

// PCG64 jump-ahead table (nsg_rng.hip.h): the pair (M^e, 1 + M + ... + M^(e-1)) mod 2^128 for the exponents e = 0 .. kJumpLow-1
// (digit-0 block), then for e = v * 256^d, d = 1 .. 4, v = 0 .. 255.
void nsg_pcg64_jump_table(uint64_t* out) {
  typedef unsigned __int128 u128;
  const u128 M = ((u128)2549297995355413924ULL << 64) | 4865540595714422341ULL;   // PCG_DEFAULT_MULTIPLIER_128
  auto put = [&](size_t idx, u128 a, u128 g) {
    uint64_t* e = out + idx * 4;
    e[0] = (uint64_t)(a >> 64); e[1] = (uint64_t)a; e[2] = (uint64_t)(g >> 64); e[3] = (uint64_t)g;
  };
  u128 base_a = 1, base_g = 0;   // exponent 256^d of the block being written
  {
    u128 a = 1, g = 0;           // exponent 0
    for (int v = 0; v < kJumpLow; v++) {
      put((size_t)v, a, g);
      if (v == 256) { base_a = a; base_g = g; }
      g = g * M + 1;             // exponents x then 1:  G_(x+1) = G_x * M + 1
      a = a * M;
    }
  }
  for (int d = 1; d < kJumpDigits; d++) {
    u128 a = 1, g = 0;
    for (int v = 0; v < 256; v++) {
      put((size_t)kJumpLow + (size_t)(d - 1) * 256 + v, a, g);
      g = g * base_a + base_g;   // exponents x then 256^d:  G_(x+y) = G_x * A_y + G_y
      a = a * base_a;
    }
    base_a = a;                  // after 256 compositions: exponent 256^(d+1)
    base_g = g;
  }
}

int nsg_abi_version(void) { return NSG_ABI_VERSION; }
const char* nsg_last_error(void) { return g_err; }
size_t nsg_sizeof_config(void) { return sizeof(nsg_config); }
size_t nsg_sizeof_buffers(void) { return sizeof(nsg_buffers); }
size_t nsg_sizeof_layout(void) { return sizeof(nsg_layout); }

int nsg_layout_query(const nsg_config* cfg, int64_t n, nsg_layout* out) {
  if (!out) return fail(NSG_EINVAL, "out is NULL");
  if (n <= 0 || n > NSG_MAX_ENVS) return fail(NSG_EINVAL, "n must be in [1, 2^27]");
  if (!cfg || cfg->env_type < 0 || cfg->env_type >= NSG_ENV_COUNT) return fail(NSG_EINVAL, "bad config");
  memset(out, 0, sizeof(*out));
  const int e = cfg->env_type, P = cfg->n_params;
  const bool fl = is_grid_env(e);
  bool any_rng = false, any_cursor = false, any_sched = false;
  for (int p = 0; p < P; p++) {
    any_rng |= cfg->params[p].uses_rng != 0;
    any_sched |= sched_is_stochastic(cfg->params[p].sched_kind);
    const int k = cfg->params[p].upd_kind;
    any_cursor |= upd_uses_cursor(k);
  }
  out->n = n;
  out->phys_dim = kPhysDim[e];
  out->obs_dim = kObsDim[e];
  out->n_params = P;
  out->n_theta_rows = fl ? kNDist[e] * P : P;
  out->action_is_float = (e == NSG_ENV_PENDULUM || e == NSG_ENV_MOUNTAINCAR_CONT) ? 1 : 0;
  out->n_actions = kNActions[e];
  out->phys = (int64_t)kPhysDim[e] * ((n + kBlock - 1) / kBlock) * kBlock;  // chunk-blocked: [ceil(n/256)][F][256]
  out->cell = fl ? n : 0;
  out->theta = (int64_t)out->n_theta_rows * n;
  out->table_prob = (fl && e != NSG_ENV_BRIDGE) ? (int64_t)kNDist[e] * ((n + kBlock - 1) / kBlock) * kBlock : 0;  // chunk-blocked
  out->t = n;
  const bool simenv = (cfg->flags & NSG_F_SIM_ENV) != 0;
  out->t_fork = simenv ? n : 0;
  out->derived = (simenv && e == NSG_ENV_CARTPOLE) ? 2 * n : (simenv && e == NSG_ENV_CLIFFWALKING) ? 4 * n : 0;
  out->status = fl ? n : 0;     // grid envs: a byte; classic-control envs: the episode word
  out->episode = fl ? 0 : n;
  // grid envs: chunk-blocked PCG64 state rows; classic-control envs: descriptor + one (seed, spawn key) record per env
  out->rng_env = fl ? 4 * ((n + kBlock - 1) / kBlock) * kBlock : 2 * (n + 1);
  out->rng_upd = any_rng ? (int64_t)P * 4 * n : 0;
  out->cursor = any_cursor ? (int64_t)P * n : 0;
  out->rng_sched = any_sched ? (int64_t)P * 4 * n : 0;
  out->sched_next = any_sched ? (int64_t)P * n : 0;
  out->obs = fl ? 0 : (int64_t)kObsDim[e] * n;
  out->reward = n;
  out->terminated = n;
  out->truncated = n;
  out->env_change = (int64_t)(P > 0 ? P : 1) * n;
  out->delta_change = (int64_t)(P > 0 ? P : 1) * n;
  out->prob = fl ? n : 0;
  out->violation = ((cfg->flags & NSG_F_VIOLATION_MASK) && !fl) ? (int64_t)(P > 0 ? P : 1) * n : 0;
  // episode accounting (NSG_F_TRACK_RETURNS): CartPole pays +1 and MountainCar -1 on EVERY step, so their episode return is
  // +-length: no running-return row and no last_return row (one scattered store per finished episode instead of two)
  const bool tr = (cfg->flags & NSG_F_TRACK_RETURNS) != 0;
  const bool ret_from_len = e == NSG_ENV_CARTPOLE || e == NSG_ENV_MOUNTAINCAR;
  out->ep_return = (tr && !ret_from_len) ? n : 0;
  out->ep_length = 0;          // the running length is the wrapper time t
  out->last_return = (tr && !ret_from_len) ? n : 0;
  out->last_length = tr ? n : 0;
  out->counters = NSG_CNT_COUNT * NSG_CNT_SHARDS;
  out->done_bits = (n + 63) / 64;
  return NSG_OK;
}

int nsg_create(const nsg_config* cfg, const void* tables, size_t table_bytes, int64_t n, nsg_handle** out) {
  if (!out) return fail(NSG_EINVAL, "out is NULL");
  *out = nullptr;
  // the kernels address a row with a 32-bit BYTE offset per lane (32-byte stream records: i * 32 < 2^32)
  if (n <= 0 || n > NSG_MAX_ENVS) return fail(NSG_EINVAL, "n must be in [1, 2^27] envs per handle (got %lld); shard larger batches over several handles", (long long)n);
  if (table_bytes && !tables) return fail(NSG_EINVAL, "tables is NULL");
  int rc = validate(cfg, table_bytes);
  if (rc) return rc;
  nsg_handle* h = new (std::nothrow) nsg_handle();   // value-initialised: every member starts at zero / null
  if (!h) return fail(NSG_ENOMEM, "out of host memory");
  h->n = n;
  h->id = g_next_handle_id++;
  // any failure below releases what was allocated so far (nsg_destroy frees the three device blocks and the handle)
#define HIP_TRY_H(expr)                                                          \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                      \
      (void)nsg_destroy(h);                                                      \
      return fail(NSG_EHIP, "%s: %s", #expr, hipGetErrorString(e_));             \
    }                                                                            \
  } while (0)
  HIP_TRY_H(hipGetDevice(&h->device));
  const size_t tb = ((table_bytes + 7) / 8) * 8 + 8;
  HIP_TRY_H(hipMalloc((void**)&h->d_tables, tb));
  HIP_TRY_H(hipMemset(h->d_tables, 0, tb));
  if (table_bytes) HIP_TRY_H(hipMemcpy(h->d_tables, tables, table_bytes, hipMemcpyHostToDevice));
  static uint64_t zig[1536 + kJumpWords];   // ziggurat tables | PCG64 jump-ahead table (one device block)
  static std::once_flag jump_once;
  std::call_once(jump_once, [] {   // filled once: handles may be created from several threads
    memcpy(zig, NSG_ZIG_KI, 2048);
    memcpy(zig + 256, NSG_ZIG_WI_BITS, 2048);
    memcpy(zig + 512, NSG_ZIG_FI_BITS, 2048);
    memcpy(zig + 768, NSG_ZIGE_KE, 2048);
    memcpy(zig + 1024, NSG_ZIGE_WE_BITS, 2048);
    memcpy(zig + 1280, NSG_ZIGE_FE_BITS, 2048);
    nsg_pcg64_jump_table(zig + 1536);
  });
  HIP_TRY_H(hipMalloc((void**)&h->d_zig, sizeof(zig)));
  HIP_TRY_H(hipMemcpy(h->d_zig, zig, sizeof(zig), hipMemcpyHostToDevice));
  HIP_TRY_H(hipMalloc((void**)&h->dev, sizeof(Segment)));
#undef HIP_TRY_H
  h->host.cfg = *cfg;
  h->host.N = n;
  h->host.tables = h->d_tables;
  h->host.zig = h->d_zig;
  h->host.jump = h->d_zig + 1536;
  h->host.table_bytes = (int32_t)table_bytes;
  h->host.uses_normal = 0;
  h->host.simple_theta = 1;
  h->host.uses_exp = 0;
  for (int p = 0; p < cfg->n_params; p++) {
    const nsg_param_cfg& pc = cfg->params[p];
    if (upd_is_normal(pc.upd_kind)) h->host.uses_normal = 1;
    if (pc.upd_kind == NSG_UPD_D_RANDOMCAT || pc.upd_kind == NSG_UPD_D_LCBOUNDED || pc.sched_kind == NSG_SCHED_MEMORYLESS) h->host.uses_exp = 1;
    if (!upd_kind_is_simple(pc.upd_kind) || sched_is_stochastic(pc.sched_kind)) h->host.simple_theta = 0;
  }
  *out = h;
  return NSG_OK;
}

// Bytes of all rows of a handle (what one step streams through the caches, give or take the rows a config never touches).
static int64_t layout_bytes(const nsg_layout& l) {
  return 8 * (l.phys + l.theta + l.table_prob + l.derived + l.rng_env + l.rng_upd + l.rng_sched + l.done_bits) +
         4 * (l.cell + l.t + l.t_fork + l.episode + l.sched_next + l.cursor + l.obs + l.reward + l.delta_change + l.prob + l.ep_return + l.ep_length +
              l.last_return + l.last_length) +
         (l.status + l.terminated + l.truncated + l.env_change + l.violation);
}
static int grid_cap() {  // tuning knob (tools/kbench.py sweeps it); default from measurements
  static int cap = 0;
  if (!cap) {
    const char* e = getenv("NSG_GRID_CAP");
    cap = e ? atoi(e) : 4096 * 256 / kBlock;   // 2^20 envs per launch round, whatever the workgroup size
    if (cap < 1 || cap > 65536) cap = NSG_CNT_SHARDS / (kBlock / 64);  // beyond 4096 workgroups the counter shards are shared (atomic adds)
  }
  return cap;
}
static int grid_for(int64_t n) {
  int64_t chunks = (n + kBlock - 1) / kBlock;
  return (int)(chunks < grid_cap() ? chunks : grid_cap());
}
// Workgroups of an nsg_step launch.  Batches of up to 2^20 envs (<= 4096 chunks) run best with 6 workgroups per CU (1536),
// every one resident from the start and walking 2-3 chunks - not with one workgroup per chunk dispatched in two rounds: C1 2^19
// envs 16.0 -> 14.1 us, 2^20 25.8 -> 25.0, Pendulum 20.9 -> 19.8, C3 19.5 -> 19.1; C2's specialised kernel, once it fitted 6
// wavefronts per SIMD, 34.8 -> 32.9.  That needs a kernel whose registers ALLOW 6 workgroups per CU: at 5 the sixth waits for
// a second round and the launch is 8 % slower than with 4096 (C1 at 81 VGPRs: 26.3 us).  So: the plain-arithmetic generic
// kernels (built at <= 80 VGPRs for the env types this matters for) and any specialised kernel whose register count says so.
// 1024, 1366, 1792, 2048 are all worse (C1 2^20: 25.5 / 27.8 / 26.1 / 25.0 vs 24.0 us, profiles/r02_ab_runs.txt); from 2^21
// envs on 4096 and more (next paragraph).  NSG_GRID_CAP overrides everything.
//
// Beyond 2^20 envs (more than 4096 chunks) a launch of a kernel that stages nothing into LDS and is bound by memory runs best
// with up to 16384 workgroups - one chunk each at 2^22 envs, 2 / 4 at 2^23 / 2^24 - rather than 4096 walking 4-16 chunks
// (specialised C1: 2^22 84.9 -> 80.8 us, 2^23 183 -> 174.5, 2^24 414.9 -> 390.6; MountainCar 55.7 -> 54.5, C3 84.1 -> 83.3,
// fused rollouts 46.0 -> 44.0 per step; 65536 is no better).  Not for the full theta-engine, whose every workgroup first stages
// the ziggurat tables (C2 2^22: 138 -> 150 us), nor for Acrobot, bound by its arithmetic (206.8 -> 209.6).  The counter shards
// (NSG_CNT_SHARDS = 16384, one per wavefront of a 4096-workgroup launch) are then shared by four wavefronts each: plain atomic adds.
static bool wide_grid_ok(const nsg_handle* h) {
  return h->host.simple_theta && !h->host.uses_normal && !h->host.uses_exp && h->host.cfg.env_type != NSG_ENV_ACROBOT;
}
static int launch_grid_for(const nsg_handle* h) {   // nsg_step beyond the 1536 policy, nsg_rollout
  static const bool cap_overridden = getenv("NSG_GRID_CAP") != nullptr;   // read once: this runs on every launch
  const int64_t chunks = (h->n + kBlock - 1) / kBlock;
  if (!cap_overridden && chunks > 4096 * 256 / kBlock && wide_grid_ok(h)) {
    const int64_t wide = 16384 * 256 / kBlock;
    return (int)(chunks < wide ? chunks : wide);
  }
  return grid_for(h->n);
}
static int step_grid_for(const nsg_handle* h) {
  const int64_t chunks = (h->n + kBlock - 1) / kBlock;
  const bool six_fit = h->spec ? (h->spec->step_waves >= 6 || (h->spec->step_waves == 0 && h->host.simple_theta)) : h->host.simple_theta;
  static const bool cap_overridden = getenv("NSG_GRID_CAP") != nullptr;
  if (!cap_overridden && six_fit && chunks > 1536 && chunks <= 4096) return 1536 * 256 / kBlock;
  return launch_grid_for(h);
}

int nsg_bind(nsg_handle* h, const nsg_buffers* bufs) {
  if (!h || !bufs) return fail(NSG_EINVAL, "NULL argument");
  nsg_layout lay;
  int rc = nsg_layout_query(&h->host.cfg, h->n, &lay);
  if (rc) return rc;
#define NEED(field) \
  if (lay.field > 0 && !bufs->field) return fail(NSG_EINVAL, "buffer '%s' is required (%lld elements)", #field, (long long)lay.field)
  NEED(phys); NEED(cell); NEED(theta); NEED(table_prob); NEED(derived); NEED(t); NEED(t_fork); NEED(status); NEED(episode); NEED(rng_env); NEED(rng_upd); NEED(rng_sched); NEED(sched_next); NEED(cursor);
  NEED(obs); NEED(reward); NEED(terminated); NEED(truncated); NEED(env_change); NEED(delta_change); NEED(violation);
  NEED(ep_return); NEED(ep_length); NEED(last_return); NEED(last_length);
#undef NEED
  h->host.buf = *bufs;
  h->generation++;
  HIP_TRY(hipMemcpy(h->dev, &h->host, sizeof(Segment), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(init_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, 0, h->dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  h->bound = true;
  return NSG_OK;
}

#define DISPATCH_ENV(env, CALL)                                                              \
  switch (env) {                                                                             \
    case NSG_ENV_CARTPOLE: { constexpr int E = NSG_ENV_CARTPOLE; CALL; } break;              \
    case NSG_ENV_PENDULUM: { constexpr int E = NSG_ENV_PENDULUM; CALL; } break;              \
    case NSG_ENV_ACROBOT: { constexpr int E = NSG_ENV_ACROBOT; CALL; } break;                \
    case NSG_ENV_MOUNTAINCAR: { constexpr int E = NSG_ENV_MOUNTAINCAR; CALL; } break;        \
    case NSG_ENV_MOUNTAINCAR_CONT: { constexpr int E = NSG_ENV_MOUNTAINCAR_CONT; CALL; } break; \
    case NSG_ENV_FROZENLAKE: { constexpr int E = NSG_ENV_FROZENLAKE; CALL; } break;          \
    case NSG_ENV_CLIFFWALKING: { constexpr int E = NSG_ENV_CLIFFWALKING; CALL; } break;      \
    default: { constexpr int E = NSG_ENV_BRIDGE; CALL; } break;                              \
  }

int nsg_reset(nsg_handle* h, const uint64_t* seeds_dev, const uint8_t* mask_dev, void* stream) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (!h->bound) return fail(NSG_ENOTBOUND, "nsg_bind() has not been called");
  hipStream_t s = (hipStream_t)stream;
  if (seeds_dev && !is_grid_env(h->host.cfg.env_type)) {
    // arbitrary per-env seeds: the classic-control streams leave their affine form first (stream-ordered: records, then the
    // descriptor, then the reset that writes the seeded envs' records)
    hipLaunchKernelGGL(materialize_streams_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, s, h->dev);
    hipLaunchKernelGGL(stream_set_kernel, dim3(1), dim3(1), 0, s, h->dev, 0ULL, 0ULL);
  }
  DISPATCH_ENV(h->host.cfg.env_type,
               hipLaunchKernelGGL(reset_kernel<E>, dim3(grid_for(h->n)), dim3(kBlock), 0, s, h->dev, seeds_dev, mask_dev, 0, (uint64_t)0));
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

// Fills seeds[i] = base + i (grid envs take their seeds as an array)
__global__ void iota_seeds_kernel(uint64_t* out, uint64_t base, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = base + (uint64_t)i;
}

int nsg_reset_seeded(nsg_handle* h, uint64_t base_seed, void* stream) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (!h->bound) return fail(NSG_ENOTBOUND, "nsg_bind() has not been called");
  hipStream_t s = (hipStream_t)stream;
  if (is_grid_env(h->host.cfg.env_type)) {   // grid envs keep per-env PCG64 rows: seed them from base + i
    uint64_t* seeds = nullptr;
    HIP_TRY(hipMallocAsync((void**)&seeds, sizeof(uint64_t) * (size_t)h->n, s));
    hipLaunchKernelGGL(iota_seeds_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, s, seeds, base_seed, h->n);
    DISPATCH_ENV(h->host.cfg.env_type,
                 hipLaunchKernelGGL(reset_kernel<E>, dim3(grid_for(h->n)), dim3(kBlock), 0, s, h->dev, (const uint64_t*)seeds, (const uint8_t*)nullptr, 0, (uint64_t)0));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipFreeAsync(seeds, s));
    return NSG_OK;
  }
  // classic-control envs: env i <- PCG64(SeedSequence(base + i)), kept in the affine form (one descriptor, nothing per env)
  hipLaunchKernelGGL(stream_set_kernel, dim3(1), dim3(1), 0, s, h->dev, NSG_STREAM_AFFINE | 0xffffffffULL, base_seed);
  DISPATCH_ENV(h->host.cfg.env_type,
               hipLaunchKernelGGL(reset_kernel<E>, dim3(grid_for(h->n)), dim3(kBlock), 0, s, h->dev, (const uint64_t*)nullptr, (const uint8_t*)nullptr, 1, base_seed));
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

// Every other step launch of a handle walks its chunks back to front (step_body: it starts with the rows the previous
// launch wrote last).  NSG_ALT_ORDER=0 keeps every launch front to back.
static int next_traversal(nsg_handle* h) {
  static const bool alternate = [] { const char* e = getenv("NSG_ALT_ORDER"); return !(e && e[0] == '0'); }();
  return alternate ? (int)(h->launches++ & 1u) : 0;
}

// NSG_F_LIBM_EXACT lives in the specialised units only (the precompiled kernels carry the fast sincos): a launch that would fall back to
// them is refused instead of silently stepping in the other arithmetic.
static int exact_needs_unit(const nsg_handle* h, const void* unit, const char* what) {
  if ((h->host.cfg.flags & NSG_F_LIBM_EXACT) && !unit)
    return fail(NSG_EUNSUPPORTED, "%s: this handle was created with NSG_F_LIBM_EXACT, which runs on its specialised unit only "
                                  "(nsg_specialize first; it needs the runtime compiler or a prebuilt unit)", what);
  return NSG_OK;
}

int nsg_step(nsg_handle* h, const void* actions_dev, void* stream) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (!h->bound) return fail(NSG_ENOTBOUND, "nsg_bind() has not been called");
  if (!actions_dev) return fail(NSG_EINVAL, "actions_dev is NULL");
  if (int rc = exact_needs_unit(h, h->spec, "nsg_step")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int grid = step_grid_for(h);
  const size_t lds = (size_t)lds_bytes_for(h->host.table_bytes, h->host.uses_normal, h->host.uses_exp);
  int reverse = next_traversal(h);
  if (h->spec) {
    void* args[] = {(void*)&h->dev, (void*)&actions_dev, (void*)&reverse};
    HIP_TRY(hipModuleLaunchKernel(h->spec->step, grid, 1, 1, kBlock, 1, 1, (unsigned)lds, s, args, nullptr));
  } else if (h->host.simple_theta) {
    DISPATCH_ENV(h->host.cfg.env_type,
                 hipLaunchKernelGGL((step_kernel<E, false>), dim3(grid), dim3(kBlock), lds, s, h->dev, actions_dev, reverse));
  } else {
    DISPATCH_ENV(h->host.cfg.env_type,
                 hipLaunchKernelGGL((step_kernel<E, true>), dim3(grid), dim3(kBlock), lds, s, h->dev, actions_dev, reverse));
  }
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

// the last step of a fused rollout landed in the handle's own output rows: mirror them into the last trajectory slice
static int mirror_last_slice(nsg_handle* h, const nsg_rollout_out& o, int32_t k_steps, hipStream_t s) {
  const nsg_buffers& bb = h->host.buf;
  const int64_t n = h->n, K1 = k_steps - 1;
  const int e = h->host.cfg.env_type;
  const int P = h->host.cfg.n_params > 0 ? h->host.cfg.n_params : 1;
  if (o.obs) {
    if (is_grid_env(e)) HIP_TRY(hipMemcpyAsync((int32_t*)o.obs + K1 * n, bb.cell, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemcpyAsync(o.obs + K1 * n * kObsDim[e], bb.obs, n * kObsDim[e] * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  if (o.reward) HIP_TRY(hipMemcpyAsync(o.reward + K1 * n, bb.reward, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (o.terminated) HIP_TRY(hipMemcpyAsync(o.terminated + K1 * n, bb.terminated, n, hipMemcpyDeviceToDevice, s));
  if (o.truncated) HIP_TRY(hipMemcpyAsync(o.truncated + K1 * n, bb.truncated, n, hipMemcpyDeviceToDevice, s));
  if (o.env_change) HIP_TRY(hipMemcpyAsync(o.env_change + K1 * P * n, bb.env_change, (size_t)P * n, hipMemcpyDeviceToDevice, s));
  if (o.delta_change) HIP_TRY(hipMemcpyAsync(o.delta_change + K1 * P * n, bb.delta_change, (size_t)P * n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return NSG_OK;
}

int nsg_rollout(nsg_handle* h, const void* actions_dev, int32_t k_steps, const nsg_rollout_out* out, void* stream) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (!h->bound) return fail(NSG_ENOTBOUND, "nsg_bind() has not been called");
  if (!actions_dev || k_steps <= 0) return fail(NSG_EINVAL, "bad rollout arguments");
  if (int rc = exact_needs_unit(h, h->spec, "nsg_rollout")) return rc;
  nsg_rollout_out o;
  memset(&o, 0, sizeof(o));
  if (out) o = *out;
  hipStream_t s = (hipStream_t)stream;
  // fused rollouts of the classic envs keep the chunk's env PCG64 records in LDS
  const size_t rollout_lds = (size_t)lds_bytes_for(h->host.table_bytes, h->host.uses_normal, h->host.uses_exp) +
                             (is_grid_env(h->host.cfg.env_type) ? 0 : kLdsStreamBytes * (1 + upd_lds_count(h->host.cfg)));
  if (h->spec) {
    void* args[] = {(void*)&h->dev, (void*)&actions_dev, (void*)&k_steps, (void*)&o};
    HIP_TRY(hipModuleLaunchKernel(h->spec->rollout, launch_grid_for(h), 1, 1, kBlock, 1, 1, (unsigned)rollout_lds, s, args, nullptr));
  } else if (h->host.simple_theta) {
    DISPATCH_ENV(h->host.cfg.env_type,
                 hipLaunchKernelGGL((rollout_kernel<E, false>), dim3(launch_grid_for(h)), dim3(kBlock), rollout_lds, s, h->dev, actions_dev, k_steps, o));
  } else {
    DISPATCH_ENV(h->host.cfg.env_type,
                 hipLaunchKernelGGL((rollout_kernel<E, true>), dim3(launch_grid_for(h)), dim3(kBlock), rollout_lds, s, h->dev, actions_dev, k_steps, o));
  }
  HIP_TRY(hipGetLastError());
  return mirror_last_slice(h, o, k_steps, s);
}

// ---- fused policy rollouts (nsg_rollout.hip.h: rollout_body<ENV, FULL, true>) ---------------------------------------------------
static const float kActLow[NSG_ENV_COUNT] = {0.f, -2.f, 0.f, 0.f, -1.f, 0.f, 0.f, 0.f};   // Pendulum max_torque 2.0, MountainCarContinuous +-1 [UPSTREAM]
static const float kActHigh[NSG_ENV_COUNT] = {0.f, 2.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f};

uint64_t nsg_policy_bits(uint64_t seed, uint64_t env_index, uint64_t step) { return pol_bits(seed, env_index, step); }
// key of the policy unit of action source `kind` for the handle whose main unit has key `h0`
static uint64_t policy_unit_key(uint64_t h0, int kind) {
  const char tag[8] = {'p', 'o', 'l', 'i', 'c', 'y', (char)('0' + kind), 0};
  return nsg_spec::fnv1a(tag, 7, h0);
}

int nsg_rollout_policy(nsg_handle* h, const nsg_policy* pol, int32_t k_steps, const nsg_rollout_out* out, const nsg_episode_acc* acc, void* stream) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (!h->bound) return fail(NSG_ENOTBOUND, "nsg_bind() has not been called");
  if (!pol || k_steps <= 0) return fail(NSG_EINVAL, "bad policy-rollout arguments");
  const int e = h->host.cfg.env_type;
  const bool grid_env = is_grid_env(e);
  switch (pol->kind) {
    case NSG_POL_TABLE:
      if (!pol->data) return fail(NSG_EINVAL, "NSG_POL_TABLE: data (actions[K][N]) is NULL");
      break;
    case NSG_POL_UNIFORM:
      if (pol->step0 < 0 || pol->index0 < 0) return fail(NSG_EINVAL, "NSG_POL_UNIFORM: step0 and index0 must not be negative");
      break;
    case NSG_POL_BY_STATE:
      if (!grid_env) return fail(NSG_EINVAL, "NSG_POL_BY_STATE needs a discrete state: grid envs only (classic control: NSG_POL_LINEAR)");
      if (!pol->data || pol->n_data < h->host.cfg.nrow * h->host.cfg.ncol)
        return fail(NSG_EINVAL, "NSG_POL_BY_STATE: the table must hold one action per cell (%d), got %d", h->host.cfg.nrow * h->host.cfg.ncol, pol->n_data);
      break;
    case NSG_POL_LINEAR:
      if (grid_env) return fail(NSG_EINVAL, "NSG_POL_LINEAR works on the float32 observation of the classic-control envs (grid envs: NSG_POL_BY_STATE)");
      if (!pol->data || pol->n_data != (kNActions[e] > 0 ? kNActions[e] : 1))
        return fail(NSG_EINVAL, "NSG_POL_LINEAR: %d weight rows of obs_dim + 1 floats are required, got %d", kNActions[e] > 0 ? kNActions[e] : 1, pol->n_data);
      break;
    default: return fail(NSG_EINVAL, "unknown policy kind %d", pol->kind);
  }
  PolicyArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.pol = *pol;
  if (acc) pa.acc = *acc;
  if (pa.acc.discount && pa.acc.n_discount <= 0) pa.acc.discount = nullptr;
  pa.act_lo = kActLow[e];
  pa.act_hi = kActHigh[e];
  pa.n_actions = kNActions[e];
  nsg_rollout_out o;
  memset(&o, 0, sizeof(o));
  if (out) o = *out;
  hipStream_t s = (hipStream_t)stream;
  const size_t rollout_lds = (size_t)lds_bytes_for(h->host.table_bytes, h->host.uses_normal, h->host.uses_exp) +
                             (grid_env ? 0 : kLdsStreamBytes * (1 + upd_lds_count(h->host.cfg)));
  const int grid = launch_grid_for(h);
  bool launched = false;
  if (h->spec) {   // a specialised handle: the policy rollout compiled for its configuration (its own unit, built on first use)
    const int kind = pol->kind;
    if (!h->spec_policy_unit[kind] && !h->spec_policy_tried[kind]) {
      h->spec_policy_tried[kind] = true;
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, h->device));
      const SpecPolicy sp = spec_policy(h->host.cfg, h->n);
      const int grc = get_spec_module(h->device, policy_unit_key(h->spec->h0, kind), kUnitPolicy,
                                      [&](std::string& err) { return nsg_spec::policy_compile(h->host.cfg, sp.full, kind, prop.gcnArchName, err, sp.inlane, sp.stream_state); },
                                      &h->spec_policy_unit[kind]);
      if (grc) h->spec_policy_unit[kind] = nullptr;   // the generic kernel stays in force
    }
    if (h->spec_policy_unit[kind]) {
      void* args[] = {(void*)&h->dev, (void*)&k_steps, (void*)&o, (void*)&pa};
      HIP_TRY(hipModuleLaunchKernel(h->spec_policy_unit[kind]->rollout_policy, grid, 1, 1, kBlock, 1, 1, (unsigned)rollout_lds, s, args, nullptr));
      launched = true;
    }
  }
  if (!launched) {
    if (int rc = exact_needs_unit(h, nullptr, "nsg_rollout_policy")) return rc;
    if (h->host.simple_theta) {
      DISPATCH_ENV(e, hipLaunchKernelGGL((rollout_policy_kernel<E, false>), dim3(grid), dim3(kBlock), rollout_lds, s, h->dev, k_steps, o, pa));
    } else {
      DISPATCH_ENV(e, hipLaunchKernelGGL((rollout_policy_kernel<E, true>), dim3(grid), dim3(kBlock), rollout_lds, s, h->dev, k_steps, o, pa));
    }
  }
  HIP_TRY(hipGetLastError());
  return mirror_last_slice(h, o, k_steps, s);
}
/* Which kernel nsg_rollout_policy launches for this handle: 0 generic, 1 the specialised unit (after the first such rollout). */
int nsg_rollout_policy_kind(const nsg_handle* h, int32_t kind) { return h && kind >= 0 && kind < 4 && h->spec_policy_unit[kind] ? 1 : 0; }

// ---- resident stepper (nsg_rollout.hip.h: resident_body) -----------------------------------------------------------------
// The waits of the resident kernels are budgets of the device's steady counter (wall_clock64()).  Its rate is MEASURED, once per
// device: a one-lane kernel spins until the counter has advanced by 2^20 ticks between two events (10.49 ms on the MI355X: 100 MHz).
// Measured rather than assumed: a first cut read s_memrealtime and assumed 100 MHz - it left after 0.17 ms of a 5-ms budget.
__global__ void wall_clock_spin_kernel(uint64_t ticks, uint64_t* out) {
  const uint64_t t0 = (uint64_t)wall_clock64();
  uint64_t t = t0;
  for (int guard = 0; guard < (1 << 28) && t - t0 < ticks; guard++) {   // (the guard bounds the spin whatever the counter does)
    __builtin_amdgcn_s_sleep(8);
    t = (uint64_t)wall_clock64();
  }
  if (out) *out = t - t0;
}
static int wall_ticks_per_us(int device, double* out) {
  static std::mutex m;
  static std::map<int, double> rate;
  std::lock_guard<std::mutex> lock(m);
  auto it = rate.find(device);
  if (it == rate.end()) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    const uint64_t ticks = 1u << 20;
    HIP_TRY(hipEventCreate(&e0));
    hipError_t he = hipEventCreate(&e1);
    if (he == hipSuccess) { hipLaunchKernelGGL(wall_clock_spin_kernel, dim3(1), dim3(1), 0, 0, (uint64_t)1, (uint64_t*)nullptr); he = hipDeviceSynchronize(); }   // warm
    if (he == hipSuccess) he = hipEventRecord(e0, 0);
    if (he == hipSuccess) { hipLaunchKernelGGL(wall_clock_spin_kernel, dim3(1), dim3(1), 0, 0, ticks, (uint64_t*)nullptr); he = hipGetLastError(); }
    if (he == hipSuccess) he = hipEventRecord(e1, 0);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(NSG_EHIP, "measuring the device's wall-clock rate: %s", hipGetErrorString(he));
    if (!(ms > 0.01f)) return fail(NSG_EHIP, "the device's wall clock does not advance: the resident stepper cannot bound its waits");
    it = rate.emplace(device, (double)ticks / ((double)ms * 1000.0)).first;
  }
  *out = it->second;
  return NSG_OK;
}

int nsg_resident_start(nsg_handle* h, const void* actions_dev, nsg_mailbox* mb_dev, int32_t max_steps, uint32_t wait_budget_us, void* stream) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (!h->bound) return fail(NSG_ENOTBOUND, "nsg_bind() has not been called");
  if (!actions_dev || !mb_dev || max_steps <= 0 || wait_budget_us == 0) return fail(NSG_EINVAL, "bad resident-stepper arguments");
  if (h->n > NSG_RESIDENT_MAX_ENVS)
    return fail(NSG_EINVAL, "the resident stepper keeps one workgroup per 256-env chunk on the device at once: at most %d envs (this batch: %lld); "
                            "larger batches are bound by memory, not by launches - use nsg_step / nsg_rollout", NSG_RESIDENT_MAX_ENVS, (long long)h->n);
  const int grid = (int)((h->n + kBlock - 1) / kBlock);
  const size_t lds = (size_t)lds_bytes_for(h->host.table_bytes, h->host.uses_normal, h->host.uses_exp) +
                     (is_grid_env(h->host.cfg.env_type) ? 0 : kLdsStreamBytes * (1 + upd_lds_count(h->host.cfg)));
  ResidentArgs ra;
  ra.mb = mb_dev;
  ra.max_steps = max_steps;
  ra.reserved = 0;
  double per_us = 0.0;   // the device's steady wall clock (wall_clock64() in the kernel)
  int rc = wall_ticks_per_us(h->device, &per_us);
  if (rc) return rc;
  ra.budget_ticks = (uint64_t)((double)wait_budget_us * per_us);
  ra.grace_ticks = (uint64_t)((double)NSG_RESIDENT_GRACE_US * per_us);
  hipStream_t s = (hipStream_t)stream;
  if (h->spec) {   // a specialised handle: the resident kernel compiled for its configuration (its own unit, built on first use)
    if (!h->spec_resident) {
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, h->device));
      const bool full = !h->host.simple_theta;
      uint64_t h0 = h->spec->h0;
      h0 = nsg_spec::fnv1a("resident", 8, h0);
      const int grc = get_spec_module(h->device, h0, kUnitResident,
                                      [&](std::string& err) { return nsg_spec::resident_compile(h->host.cfg, full, prop.gcnArchName, err); }, &h->spec_resident);
      if (grc) h->spec_resident = nullptr;   // the generic resident kernel stays in force
    }
    if (h->spec_resident) {
      void* args[] = {(void*)&h->dev, (void*)&actions_dev, (void*)&ra};
      HIP_TRY(hipModuleLaunchKernel(h->spec_resident->resident, grid, 1, 1, kBlock, 1, 1, (unsigned)lds, s, args, nullptr));
      return NSG_OK;
    }
  }
  if (int rc = exact_needs_unit(h, nullptr, "nsg_resident_start")) return rc;
  if (h->host.simple_theta) {
    DISPATCH_ENV(h->host.cfg.env_type, hipLaunchKernelGGL((resident_kernel<E, false>), dim3(grid), dim3(kBlock), lds, s, h->dev, actions_dev, ra));
  } else {
    DISPATCH_ENV(h->host.cfg.env_type, hipLaunchKernelGGL((resident_kernel<E, true>), dim3(grid), dim3(kBlock), lds, s, h->dev, actions_dev, ra));
  }
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

int nsg_resident_publish(nsg_handle* h, nsg_mailbox* mb_dev, int32_t step, void* stream) {
  if (!h || !mb_dev || step < 0) return fail(NSG_EINVAL, "bad arguments");
  const int chunks = (int)((h->n + kBlock - 1) / kBlock);
  if (chunks > NSG_RESIDENT_MAX_CHUNKS) return fail(NSG_EINVAL, "at most %d envs", NSG_RESIDENT_MAX_ENVS);
  hipLaunchKernelGGL(resident_publish_kernel, dim3((chunks + 255) / 256), dim3(256), 0, (hipStream_t)stream, mb_dev, chunks, (uint64_t)step + 1u);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

int nsg_resident_demo_policy(nsg_handle* h, int32_t watch, int32_t* actions_dev, nsg_mailbox* mb_dev, int32_t max_steps, uint32_t wait_budget_us,
                             void* stream) {
  if (!h || !h->bound) return fail(NSG_ENOTBOUND, "a bound handle is required");
  const int e = h->host.cfg.env_type;
  if (is_grid_env(e) || kNActions[e] <= 0) return fail(NSG_EINVAL, "the demo policy drives discrete-action classic-control envs");
  if (!actions_dev || !mb_dev || max_steps <= 0 || wait_budget_us == 0 || watch < 0 || watch >= kObsDim[e]) return fail(NSG_EINVAL, "bad demo-policy arguments");
  if (h->n > NSG_RESIDENT_MAX_ENVS) return fail(NSG_EINVAL, "at most %d envs", NSG_RESIDENT_MAX_ENVS);
  ResidentArgs ra;
  ra.mb = mb_dev;
  ra.max_steps = max_steps;
  ra.reserved = 0;
  double per_us = 0.0;
  int rc = wall_ticks_per_us(h->device, &per_us);
  if (rc) return rc;
  ra.budget_ticks = (uint64_t)((double)wait_budget_us * per_us);
  ra.grace_ticks = (uint64_t)((double)NSG_RESIDENT_GRACE_US * per_us);
  const int grid = (int)((h->n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(resident_demo_policy_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (const float*)h->host.buf.obs, kObsDim[e], watch,
                     actions_dev, h->n, kNActions[e], ra);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

// ---- heterogeneous launches: plans ---------------------------------------------------------------------------------------
// A launch of nsg_step_group reads a segment table: copies of its members' segments with their block ranges, passed as a
// `const __restrict__` kernel argument (scalar loads, see step_group_kernel).  What is remembered about a member list is a PLAN:
// the table, the launch shape and - when every member is specialised - the group's own unit.  Plans are process-wide (a mutex
// guards the list; the launch itself is enqueued outside it), keyed on the members' ids and valid for the members' own
// generations, so planning copies created or destroyed elsewhere in the process never invalidate a group they are not part of.
// A table is written ONCE, before the first launch that reads it, and never overwritten: a re-plan (a member re-bound or
// specialised since) allocates a new table and RETIRES the old one, which stays allocated - a launch in flight or a captured
// HIP graph may still read it - until the retired list is drained behind a device synchronisation.  Plans whose member was
// destroyed are dropped by nsg_destroy.  While the stream is capturing, a launch that would have to plan first is refused
// (planning synchronises and copies): launch the group once before the capture.
struct GroupPlan {
  uint64_t ids[NSG_MAX_SEGMENTS];
  uint64_t gens[NSG_MAX_SEGMENTS];
  int n_members = 0;
  Segment* d_table = nullptr;
  int total_blocks = 0, all_simple = 0, group_lds = 0, group_rollout_lds = 0, device = -1;
  const nsg_spec::Module* group_spec = nullptr;
  uint64_t last_used = 0;
};
constexpr size_t kMaxGroupPlans = 16;     // least recently used beyond this
constexpr size_t kMaxRetiredTables = 1024;  // 16 KB each; drained (device synchronisation + free) beyond this: a captured graph that still
                                            // points at a table outlives at most this many later re-plans / evictions (nsgym_hip.h)
static std::mutex g_plan_mutex;
static std::vector<GroupPlan> g_plans;
static std::vector<std::pair<int, Segment*>> g_retired;   // (device, table)
static uint64_t g_plan_clock = 0;

static void retire_table_locked(GroupPlan& p) {
  if (p.d_table) g_retired.emplace_back(p.device, p.d_table);
  p.d_table = nullptr;
}
// Frees the retired tables, each behind a synchronisation of ITS device (a process may hold handles on several GPUs; the current
// device is restored).  The caller has established that no stream of this thread is capturing (acquire_group_plan).
static int drain_retired_locked() {
  int cur = -1;
  (void)hipGetDevice(&cur);
  std::sort(g_retired.begin(), g_retired.end());
  int synced = -1;
  hipError_t bad = hipSuccess;
  for (auto& r : g_retired) {
    if (r.first != synced) {
      hipError_t e = hipSetDevice(r.first);
      if (e == hipSuccess) e = hipDeviceSynchronize();
      if (e != hipSuccess) { bad = e; (void)hipGetLastError(); synced = -1; continue; }   // its tables stay parked
      synced = r.first;
    }
    (void)hipFree(r.second);
    r.second = nullptr;
  }
  g_retired.erase(std::remove_if(g_retired.begin(), g_retired.end(), [](const std::pair<int, Segment*>& r) { return r.second == nullptr; }), g_retired.end());
  if (cur >= 0) (void)hipSetDevice(cur);
  return bad == hipSuccess ? NSG_OK : fail(NSG_EHIP, "draining retired segment tables: %s", hipGetErrorString(bad));
}
static bool same_members(const GroupPlan& p, nsg_handle* const* hs, int n) {
  if (p.n_members != n) return false;
  for (int k = 0; k < n; k++)
    if (p.ids[k] != hs[k]->id) return false;
  return true;
}
static bool plan_is_current(const GroupPlan& p, nsg_handle* const* hs) {
  for (int k = 0; k < p.n_members; k++)
    if (p.gens[k] != hs[k]->generation) return false;
  return true;
}

// (Re-)plans `plan` for the member list; called with g_plan_mutex held, never during a stream capture.
static int make_group_plan_locked(GroupPlan& plan, nsg_handle* const* hs, int n_handles) {
  int order[NSG_MAX_SEGMENTS];
  for (int k = 0; k < n_handles; k++) order[k] = k;
  // workgroups are dispatched in block order: the members with the longest-running workgroups get the
  // lowest block ranges (Acrobot's RK4 step takes ~3x a Pendulum step), the short ones fill in behind them
  auto cost = [&](int k) {
    static const int kEnvCost[NSG_ENV_COUNT] = {3, 2, 8, 1, 1, 2, 2, 2};  // relative time per workgroup
    return kEnvCost[hs[k]->host.cfg.env_type] + (hs[k]->host.simple_theta ? 0 : 2);
  };
  static const bool shortest_first = [] { const char* e = getenv("NSG_GROUP_ORDER"); return e && e[0] == 's'; }();
  for (int a = 1; a < n_handles; a++)
    for (int b = a; b > 0 && (shortest_first ? cost(order[b]) < cost(order[b - 1]) : cost(order[b]) > cost(order[b - 1])); b--) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  Segment tmp[NSG_MAX_SEGMENTS];
  int begin = 0;
  for (int k = 0; k < n_handles; k++) tmp[k] = hs[k]->host;
  for (int j = 0; j < n_handles; j++) {
    const int k = order[j];
    tmp[k].block_begin = begin;
    tmp[k].block_count = grid_for(hs[k]->n);
    begin += tmp[k].block_count;
  }
  retire_table_locked(plan);   // never overwritten: something in flight, or a captured graph, may still read it
  if (g_retired.size() > kMaxRetiredTables) {
    const int rc = drain_retired_locked();
    if (rc) return rc;
  }
  HIP_TRY(hipMalloc((void**)&plan.d_table, sizeof(Segment) * NSG_MAX_SEGMENTS));
  HIP_TRY(hipMemcpy(plan.d_table, tmp, sizeof(Segment) * n_handles, hipMemcpyHostToDevice));
  plan.n_members = n_handles;
  plan.total_blocks = begin;
  plan.device = hs[0]->device;
  plan.all_simple = 1;
  plan.group_lds = plan.group_rollout_lds = 0;
  bool all_spec = true;
  for (int k = 0; k < n_handles; k++) {
    plan.ids[k] = hs[k]->id;
    plan.gens[k] = hs[k]->generation;
    plan.all_simple &= hs[k]->host.simple_theta;
    all_spec = all_spec && hs[k]->spec != nullptr;
    const int l = lds_bytes_for(hs[k]->host.table_bytes, hs[k]->host.uses_normal, hs[k]->host.uses_exp);
    if (l > plan.group_lds) plan.group_lds = l;
    // a fused rollout keeps the chunk's env streams (and its first stochastic update fns' streams) in LDS as well (nsg_rollout)
    const int lr = l + (is_grid_env(hs[k]->host.cfg.env_type) ? 0 : kLdsStreamBytes * (1 + upd_lds_count(hs[k]->host.cfg)));
    if (lr > plan.group_rollout_lds) plan.group_rollout_lds = lr;
  }
  // every member runs config-specialised kernels: so does the group (one unit for the ordered tuple of configs)
  plan.group_spec = nullptr;
  if (all_spec) {
    const nsg_config* cfgs[NSG_MAX_SEGMENTS];
    bool full[NSG_MAX_SEGMENTS];
    uint64_t keys[NSG_MAX_SEGMENTS];
    for (int k = 0; k < n_handles; k++) {
      keys[k] = hs[k]->spec->h0;
      cfgs[k] = &hs[k]->host.cfg;
      full[k] = !hs[k]->host.simple_theta;
    }
    const uint64_t h0 = group_key(keys, n_handles);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, hs[0]->device));
    const int rc = get_spec_module(hs[0]->device, h0, kUnitGroup,
                                   [&](std::string& err) { return nsg_spec::group_compile(cfgs, full, n_handles, prop.gcnArchName, err); },
                                   &plan.group_spec);
    if (rc) plan.group_spec = nullptr;  // the generic group kernel stays in force
  }
  return NSG_OK;
}

static bool group_is_exact(nsg_handle* const* hs, int n_handles) {
  for (int k = 0; k < n_handles; k++)
    if (hs[k]->host.cfg.flags & NSG_F_LIBM_EXACT) return true;
  return false;
}

// An exact group runs on its own unit or not at all (the generic group kernels carry the fast sincos).
static int exact_group_needs_unit(nsg_handle* const* hs, int n_handles, const void* kernel, const char* what) {
  if (kernel || !group_is_exact(hs, n_handles)) return NSG_OK;
  return fail(NSG_EUNSUPPORTED, "%s: the members were created with NSG_F_LIBM_EXACT and no specialised unit could be built for this member list "
                                "(every member specialised? runtime compiler available? a unit that spills is refused) - the generic group kernels "
                                "carry the fast arithmetic only", what);
}

static int check_group_members(nsg_handle* const* hs, int32_t n_handles) {
  if (!hs || n_handles <= 0 || n_handles > NSG_MAX_SEGMENTS) return fail(NSG_EINVAL, "bad group arguments");
  for (int k = 0; k < n_handles; k++) {
    if (!hs[k] || !hs[k]->bound) return fail(NSG_ENOTBOUND, "group member %d is not bound", k);
    if (hs[k]->device != hs[0]->device) return fail(NSG_EINVAL, "group member %d lives on device %d, member 0 on device %d", k, hs[k]->device, hs[0]->device);
    for (int j = 0; j < k; j++)
      if (hs[j] == hs[k]) return fail(NSG_EINVAL, "group member %d is listed twice", k);
  }
  // NSG_F_LIBM_EXACT is a property of a whole unit: the classic-control members of one launch are all exact or none is (a member that
  // silently stepped in the other arithmetic would void what the flag promises); grid members have nothing to choose
  int exact = 0, fast = 0;
  for (int k = 0; k < n_handles; k++) {
    if (is_grid_env(hs[k]->host.cfg.env_type)) continue;
    if (hs[k]->host.cfg.flags & NSG_F_LIBM_EXACT) exact++; else fast++;
  }
  if (exact && fast)
    return fail(NSG_EUNSUPPORTED, "%d classic-control member(s) of this group were created with NSG_F_LIBM_EXACT and %d without: one launch runs one "
                                  "arithmetic - create all of them with the flag, or step the exact ones on their own units", exact, fast);
  return NSG_OK;
}

// What a launch needs from its plan, copied out under the mutex (the launch itself is enqueued outside it).
struct PlanSnapshot {
  const Segment* table = nullptr;
  const nsg_spec::Module* group_spec = nullptr;
  int total_blocks = 0, all_simple = 0, group_lds = 0, group_rollout_lds = 0;
};

// Finds the current plan of the member list, or makes it (never while `stream` is capturing).
static int acquire_group_plan(nsg_handle* const* hs, int32_t n_handles, void* stream, PlanSnapshot* out) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  GroupPlan* plan = nullptr;
  for (GroupPlan& p : g_plans)
    if (same_members(p, hs, n_handles)) { plan = &p; break; }
  if (!plan || !plan_is_current(*plan, hs)) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
      return fail(NSG_EINVAL, "nsg_step_group: this member list has no current plan and the stream is capturing (planning "
                              "synchronises the device): launch the group once before the capture");
    (void)hipGetLastError();
    if (!plan) {
      if (g_plans.size() >= kMaxGroupPlans) {   // least recently used goes; its table is retired, not freed
        size_t lru = 0;
        for (size_t q = 1; q < g_plans.size(); q++)
          if (g_plans[q].last_used < g_plans[lru].last_used) lru = q;
        retire_table_locked(g_plans[lru]);
        g_plans.erase(g_plans.begin() + (long)lru);
      }
      g_plans.emplace_back();
      plan = &g_plans.back();
    }
    const int rc = make_group_plan_locked(*plan, hs, n_handles);
    if (rc) {   // nothing half-made stays behind
      retire_table_locked(*plan);
      g_plans.erase(g_plans.begin() + (plan - g_plans.data()));
      return rc;
    }
  }
  plan->last_used = ++g_plan_clock;
  out->table = plan->d_table;
  out->group_spec = plan->group_spec;
  out->total_blocks = plan->total_blocks;
  out->all_simple = plan->all_simple;
  out->group_lds = plan->group_lds;
  out->group_rollout_lds = plan->group_rollout_lds;
  return NSG_OK;
}

int nsg_step_group(nsg_handle* const* hs, int32_t n_handles, const void* const* actions_dev, void* stream) {
  if (!actions_dev) return fail(NSG_EINVAL, "bad group arguments");
  int rc = check_group_members(hs, n_handles);
  if (rc) return rc;
  ActionPtrs ap;
  memset(&ap, 0, sizeof(ap));
  for (int k = 0; k < n_handles; k++) {
    if (!actions_dev[k]) return fail(NSG_EINVAL, "actions_dev[%d] is NULL", k);
    ap.p[k] = actions_dev[k];
  }
  PlanSnapshot ps;
  rc = acquire_group_plan(hs, n_handles, stream, &ps);
  if (rc) return rc;
  int reverse = next_traversal(hs[0]);   // the members of a group alternate together
  const Segment* ga = ps.table;
  if ((rc = exact_group_needs_unit(hs, n_handles, ps.group_spec ? (const void*)ps.group_spec->group : nullptr, "nsg_step_group"))) return rc;
  if (ps.group_spec) {
    void* args[] = {(void*)&ga, (void*)&n_handles, (void*)&ap, (void*)&reverse};
    HIP_TRY(hipModuleLaunchKernel(ps.group_spec->group, ps.total_blocks, 1, 1, kBlock, 1, 1, (unsigned)ps.group_lds, (hipStream_t)stream, args, nullptr));
  } else if (ps.all_simple) hipLaunchKernelGGL(step_group_kernel<false>, dim3(ps.total_blocks), dim3(kBlock), (size_t)ps.group_lds, (hipStream_t)stream, ga, n_handles, ap, reverse);
  else hipLaunchKernelGGL(step_group_kernel<true>, dim3(ps.total_blocks), dim3(kBlock), (size_t)ps.group_lds, (hipStream_t)stream, ga, n_handles, ap, reverse);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

// K fused steps of every member in ONE launch (nsg_rollout per member, a block range each): the heterogeneous counterpart of
// nsg_rollout.  actions_dev[k]: [K][N_k]; outs[k]: member k's trajectory buffers (any pointer, or `outs` itself, may be NULL).
int nsg_rollout_group(nsg_handle* const* hs, int32_t n_handles, const void* const* actions_dev, int32_t k_steps,
                      const nsg_rollout_out* outs, void* stream) {
  if (!actions_dev || k_steps <= 0) return fail(NSG_EINVAL, "bad group rollout arguments");
  int rc = check_group_members(hs, n_handles);
  if (rc) return rc;
  ActionPtrs ap;
  RolloutOuts ro;
  memset(&ap, 0, sizeof(ap));
  memset(&ro, 0, sizeof(ro));
  for (int k = 0; k < n_handles; k++) {
    if (!actions_dev[k]) return fail(NSG_EINVAL, "actions_dev[%d] is NULL", k);
    ap.p[k] = actions_dev[k];
    if (outs) ro.o[k] = outs[k];
  }
  PlanSnapshot ps;
  rc = acquire_group_plan(hs, n_handles, stream, &ps);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const Segment* ga = ps.table;
  if ((rc = exact_group_needs_unit(hs, n_handles, ps.group_spec ? (const void*)ps.group_spec->group_rollout : nullptr, "nsg_rollout_group"))) return rc;
  if (ps.group_spec && ps.group_spec->group_rollout) {
    void* args[] = {(void*)&ga, (void*)&n_handles, (void*)&ap, (void*)&k_steps, (void*)&ro};
    HIP_TRY(hipModuleLaunchKernel(ps.group_spec->group_rollout, ps.total_blocks, 1, 1, kBlock, 1, 1, (unsigned)ps.group_rollout_lds, s, args, nullptr));
  } else if (ps.all_simple) hipLaunchKernelGGL(rollout_group_kernel<false>, dim3(ps.total_blocks), dim3(kBlock), (size_t)ps.group_rollout_lds, s, ga, n_handles, ap, k_steps, ro);
  else hipLaunchKernelGGL(rollout_group_kernel<true>, dim3(ps.total_blocks), dim3(kBlock), (size_t)ps.group_rollout_lds, s, ga, n_handles, ap, k_steps, ro);
  HIP_TRY(hipGetLastError());
  // like nsg_rollout: the last step landed in each member's own output rows - mirror them into its last trajectory slice
  for (int m = 0; m < n_handles; m++) {
    const nsg_rollout_out& o = ro.o[m];
    const nsg_buffers& bb = hs[m]->host.buf;
    const int64_t n = hs[m]->n, K1 = k_steps - 1;
    const int e = hs[m]->host.cfg.env_type;
    const int P = hs[m]->host.cfg.n_params > 0 ? hs[m]->host.cfg.n_params : 1;
    if (o.obs) {
      if (is_grid_env(e)) HIP_TRY(hipMemcpyAsync((int32_t*)o.obs + K1 * n, bb.cell, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
      else HIP_TRY(hipMemcpyAsync(o.obs + K1 * n * kObsDim[e], bb.obs, n * kObsDim[e] * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (o.reward) HIP_TRY(hipMemcpyAsync(o.reward + K1 * n, bb.reward, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (o.terminated) HIP_TRY(hipMemcpyAsync(o.terminated + K1 * n, bb.terminated, n, hipMemcpyDeviceToDevice, s));
    if (o.truncated) HIP_TRY(hipMemcpyAsync(o.truncated + K1 * n, bb.truncated, n, hipMemcpyDeviceToDevice, s));
    if (o.env_change) HIP_TRY(hipMemcpyAsync(o.env_change + K1 * P * n, bb.env_change, (size_t)P * n, hipMemcpyDeviceToDevice, s));
    if (o.delta_change) HIP_TRY(hipMemcpyAsync(o.delta_change + K1 * P * n, bb.delta_change, (size_t)P * n * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return NSG_OK;
}

// Which kernel the CURRENT plan of this member list launches: NSG_GROUP_UNPLANNED (no launch yet, or a member changed since),
// NSG_GROUP_GENERIC_SIMPLE / NSG_GROUP_GENERIC_FULL (precompiled step_group_kernel<false / true>), NSG_GROUP_SPECIALISED (the
// unit compiled for the ordered tuple of the members' configs).  Negative: error.
int nsg_step_group_kind(nsg_handle* const* hs, int32_t n_handles) {
  int rc = check_group_members(hs, n_handles);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  for (const GroupPlan& p : g_plans)
    if (same_members(p, hs, n_handles) && plan_is_current(p, hs))
      return p.group_spec ? (p.group_spec->origin == NSG_SPEC_ORIGIN_PREBUILT ? NSG_GROUP_SPECIALISED_PREBUILT : NSG_GROUP_SPECIALISED)
                          : p.all_simple ? NSG_GROUP_GENERIC_SIMPLE : NSG_GROUP_GENERIC_FULL;
  return NSG_GROUP_UNPLANNED;
}

int nsg_fork(nsg_handle* src, nsg_handle* dst, uint64_t entropy, int32_t theta_mode, void* stream) {
  if (!src || !dst) return fail(NSG_EINVAL, "NULL handle");
  if (!src->bound || !dst->bound) return fail(NSG_ENOTBOUND, "both handles must be bound");
  if (dst->n < src->n || dst->n % src->n != 0)
    return fail(NSG_EINVAL, "fork: dst must hold a whole number of copies of src (%lld vs %lld envs)", (long long)dst->n, (long long)src->n);
  if (!(dst->host.cfg.flags & NSG_F_SIM_ENV)) return fail(NSG_EINVAL, "fork: dst must be created with NSG_F_SIM_ENV");
  if (theta_mode != 0 && theta_mode != 1) return fail(NSG_EINVAL, "fork: theta_mode must be 0 or 1");
  nsg_config a = src->host.cfg, b = dst->host.cfg;
  a.flags = b.flags = 0;
  a.max_episode_steps = b.max_episode_steps = 0;  // copies of CliffWalking / Bridge are re-made with 1000 (toy_text.py:229,685)
  if (memcmp(&a, &b, sizeof(a)) != 0 || src->host.table_bytes != dst->host.table_bytes)
    return fail(NSG_EINVAL, "fork: dst was not created from the same configuration");
  hipLaunchKernelGGL(fork_kernel, dim3(grid_for(dst->n)), dim3(kBlock), 0, (hipStream_t)stream, src->dev, dst->dev, entropy, theta_mode);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

int nsg_seed_streams(nsg_handle* h, const uint64_t* seeds_dev, int32_t which, void* stream) {
  if (!h || !seeds_dev) return fail(NSG_EINVAL, "NULL argument");
  if (!h->bound) return fail(NSG_ENOTBOUND, "nsg_bind() has not been called");
  if (which != 0 && which != 1) return fail(NSG_EINVAL, "which must be 0 (env) or 1 (update fns)");
  if (which == 0 && !is_grid_env(h->host.cfg.env_type)) {   // per-env seeds: the streams leave their affine form (see nsg_reset)
    hipLaunchKernelGGL(materialize_streams_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, (hipStream_t)stream, h->dev);
    hipLaunchKernelGGL(stream_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, h->dev, 0ULL, 0ULL);
  }
  hipLaunchKernelGGL(seed_streams_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, (hipStream_t)stream, h->dev, seeds_dev, which);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

// The caller wrote buffers.table_prob itself: no status byte may go on naming a config-held distribution
__global__ void table_hint_clear_kernel(uint8_t* status, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    status[i] &= (uint8_t)NSG_ST_NEEDS_RESET;   // NSG_ST_TABLE_ROWS: read the rows
}

int nsg_table_prob_dirty(nsg_handle* h, void* stream) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (!h->bound) return fail(NSG_ENOTBOUND, "nsg_bind() has not been called");
  if (!h->host.buf.table_prob || !h->host.buf.status) return NSG_OK;   // this env type keeps no P-table rows
  hipLaunchKernelGGL(table_hint_clear_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, (hipStream_t)stream, h->host.buf.status, h->n);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

int nsg_compact_done(nsg_handle* h, int32_t* idx_out_dev, uint64_t* count_out_dev, void* stream) {
  if (!h || !idx_out_dev || !count_out_dev) return fail(NSG_EINVAL, "NULL argument");
  if (!h->bound || !h->host.buf.done_bits) return fail(NSG_ENOTBOUND, "done_bits buffer is not bound");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(count_out_dev, 0, sizeof(uint64_t), s));
  hipLaunchKernelGGL(compact_done_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, s, h->host.buf.done_bits, h->n, idx_out_dev,
                     (unsigned long long*)count_out_dev);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

int nsg_theta_trace_stateful(nsg_handle* h, int32_t p, int32_t n, int32_t t0, int32_t T, const double* theta0,
                             const nsg_trace_state* state, double* theta_out, uint8_t* fired_out, double* delta_out, void* stream) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (p < 0 || p >= h->host.cfg.n_params || n <= 0 || T <= 0) return fail(NSG_EINVAL, "bad theta_trace arguments");
  if (!theta0 || !theta_out || !fired_out || !delta_out) return fail(NSG_EINVAL, "NULL buffer");
  nsg_trace_state ts;
  memset(&ts, 0, sizeof(ts));
  if (state) ts = *state;
  const nsg_param_cfg& pc = h->host.cfg.params[p];
  if (pc.uses_rng && !ts.rng) return fail(NSG_EINVAL, "rng_state required for a stochastic update fn");
  if (ts.resume && sched_is_stochastic(pc.sched_kind) && !ts.sched_rng)
    return fail(NSG_EINVAL, "resume: sched_rng required for a stochastic scheduler");
  if (ts.resume && pc.sched_kind == NSG_SCHED_MEMORYLESS && !ts.sched_next)
    return fail(NSG_EINVAL, "resume: sched_next required for MemorylessScheduler");
  if (!h->bound) HIP_TRY(hipMemcpy(h->dev, &h->host, sizeof(Segment), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(theta_trace_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock),
                     (size_t)lds_bytes_for(h->host.table_bytes, h->host.uses_normal, h->host.uses_exp), (hipStream_t)stream, h->dev, p, n, t0, T,
                     theta0, ts.rng, theta_out, fired_out, delta_out, ts);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

int nsg_theta_trace(nsg_handle* h, int32_t p, int32_t n, int32_t t0, int32_t T, const double* theta0, uint64_t* rng_state,
                    double* theta_out, uint8_t* fired_out, double* delta_out, void* stream) {
  nsg_trace_state ts;
  memset(&ts, 0, sizeof(ts));
  ts.rng = rng_state;
  return nsg_theta_trace_stateful(h, p, n, t0, T, theta0, &ts, theta_out, fired_out, delta_out, stream);
}

int nsg_rng_fill(int32_t kind, const uint64_t* seeds_dev, int32_t n, int32_t spawn_key, int32_t count, void* out_dev,
                 uint64_t* state_out_dev, void* stream) {
  if (kind < 0 || kind > 2 || !seeds_dev || n <= 0 || count < 0 || (count > 0 && !out_dev)) return fail(NSG_EINVAL, "bad rng_fill arguments");
  static thread_local uint64_t* d_zig = nullptr;
  if (!d_zig) {
    uint64_t zig[768];
    memcpy(zig, NSG_ZIG_KI, 2048);
    memcpy(zig + 256, NSG_ZIG_WI_BITS, 2048);
    memcpy(zig + 512, NSG_ZIG_FI_BITS, 2048);
    HIP_TRY(hipMalloc((void**)&d_zig, sizeof(zig)));
    HIP_TRY(hipMemcpy(d_zig, zig, sizeof(zig), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(rng_fill_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, kind, seeds_dev, n,
                     spawn_key, count, out_dev, state_out_dev, d_zig);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

int nsg_time_steps(nsg_handle* h, const void* actions_dev, int32_t iters, void* stream, float* ms_avg) {
  if (!h || !ms_avg || iters <= 0) return fail(NSG_EINVAL, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  int rc = NSG_OK;
  hipError_t he = hipEventCreate(&e0);
  if (he == hipSuccess) he = hipEventCreate(&e1);
  if (he == hipSuccess) he = hipEventRecord(e0, s);
  for (int k = 0; he == hipSuccess && rc == NSG_OK && k < iters; k++) rc = nsg_step(h, actions_dev, stream);
  if (he == hipSuccess && rc == NSG_OK) he = hipEventRecord(e1, s);
  if (he == hipSuccess && rc == NSG_OK) he = hipEventSynchronize(e1);
  if (he == hipSuccess && rc == NSG_OK) he = hipEventElapsedTime(&ms, e0, e1);
  if (e0) (void)hipEventDestroy(e0);   // on every path
  if (e1) (void)hipEventDestroy(e1);
  if (rc) return rc;
  if (he != hipSuccess) return fail(NSG_EHIP, "nsg_time_steps: %s", hipGetErrorString(he));
  *ms_avg = ms / (float)iters;
  return NSG_OK;
}

int nsg_calib_copy_f64(const double* src_dev, double* dst_dev, int64_t n, void* stream) {
  if (!src_dev || !dst_dev || n <= 0) return fail(NSG_EINVAL, "bad calib arguments");
  hipLaunchKernelGGL(calib_copy_f64_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, src_dev, dst_dev, n);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

int nsg_read_back(const void* src_dev, void* dst_host_mapped, int64_t bytes, uint64_t seq, void* stream) {
  if (!src_dev || !dst_host_mapped || bytes <= 0 || (bytes & 15) || bytes > (1 << 20) || ((uintptr_t)src_dev & 15) || ((uintptr_t)dst_host_mapped & 15))
    return fail(NSG_EINVAL, "nsg_read_back: 16-byte aligned pointers and a multiple of 16 bytes (at most 1 MiB) are required");
  hipLaunchKernelGGL(read_back_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, (const uint4*)src_dev, (uint4*)dst_host_mapped, bytes / 16, seq);
  HIP_TRY(hipGetLastError());
  return NSG_OK;
}

/* ---- config-specialised code objects (nsg_specialize.host.h) ---------------------------------- */
int nsg_spec_build(const nsg_config* cfg, const char* arch, void** code_out, size_t* size_out) {
  if (!code_out || !size_out) return fail(NSG_EINVAL, "NULL output argument");
  *code_out = nullptr;
  *size_out = 0;
  int rc = validate(cfg, (size_t)kMaxTableBytes);  // the table blob is not part of the code object: offsets checked against the limit
  if (rc) return rc;
  bool full = false;
  for (int p = 0; p < cfg->n_params; p++)
    if (!upd_kind_is_simple(cfg->params[p].upd_kind) || sched_is_stochastic(cfg->params[p].sched_kind)) full = true;
  std::string err;
  std::vector<char> code = nsg_spec::spec_compile(*cfg, full, arch && *arch ? arch : "gfx950", err);
  if (code.empty()) return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
  void* p = malloc(code.size());
  if (!p) return fail(NSG_ENOMEM, "out of host memory");
  memcpy(p, code.data(), code.size());
  *code_out = p;
  *size_out = code.size();
  return NSG_OK;
}

int nsg_spec_build_resident(const nsg_config* cfg, const char* arch, void** code_out, size_t* size_out) {
  if (!code_out || !size_out) return fail(NSG_EINVAL, "NULL output argument");
  *code_out = nullptr;
  *size_out = 0;
  int rc = validate(cfg, (size_t)kMaxTableBytes);
  if (rc) return rc;
  std::string err;
  std::vector<char> code = nsg_spec::resident_compile(*cfg, !cfg_simple_theta(*cfg), arch && *arch ? arch : "gfx950", err);
  if (code.empty()) return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
  void* p = malloc(code.size());
  if (!p) return fail(NSG_ENOMEM, "out of host memory");
  memcpy(p, code.data(), code.size());
  *code_out = p;
  *size_out = code.size();
  return NSG_OK;
}

int nsg_spec_build_policy(const nsg_config* cfg, int32_t kind, const char* arch, void** code_out, size_t* size_out) {
  if (!code_out || !size_out) return fail(NSG_EINVAL, "NULL output argument");
  if (kind < 0 || kind > 3) return fail(NSG_EINVAL, "unknown policy kind %d", kind);
  *code_out = nullptr;
  *size_out = 0;
  int rc = validate(cfg, (size_t)kMaxTableBytes);
  if (rc) return rc;
  std::string err;
  std::vector<char> code = nsg_spec::policy_compile(*cfg, !cfg_simple_theta(*cfg), kind, arch && *arch ? arch : "gfx950", err);
  if (code.empty()) return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
  void* p = malloc(code.size());
  if (!p) return fail(NSG_ENOMEM, "out of host memory");
  memcpy(p, code.data(), code.size());
  *code_out = p;
  *size_out = code.size();
  return NSG_OK;
}

int nsg_spec_build_group(const nsg_config* const* cfgs, int32_t n, const char* arch, void** code_out, size_t* size_out) {
  if (!code_out || !size_out || !cfgs || n <= 0 || n > NSG_MAX_SEGMENTS) return fail(NSG_EINVAL, "bad arguments");
  *code_out = nullptr;
  *size_out = 0;
  bool full[NSG_MAX_SEGMENTS];
  for (int k = 0; k < n; k++) {
    int rc = validate(cfgs[k], (size_t)kMaxTableBytes);
    if (rc) return rc;
    full[k] = false;
    for (int p = 0; p < cfgs[k]->n_params; p++)
      if (!upd_kind_is_simple(cfgs[k]->params[p].upd_kind) || sched_is_stochastic(cfgs[k]->params[p].sched_kind)) full[k] = true;
  }
  std::string err;
  std::vector<char> code = nsg_spec::group_compile(cfgs, full, n, arch && *arch ? arch : "gfx950", err);
  if (code.empty()) return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
  void* p = malloc(code.size());
  if (!p) return fail(NSG_ENOMEM, "out of host memory");
  memcpy(p, code.data(), code.size());
  *code_out = p;
  *size_out = code.size();
  return NSG_OK;
}

void nsg_spec_free(void* code) { free(code); }

int nsg_specialize(nsg_handle* h) {
  if (!h) return fail(NSG_EINVAL, "handle is NULL");
  if (h->spec) return NSG_OK;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, h->device));
  const SpecPolicy pol = spec_policy(h->host.cfg, h->n);
  const uint64_t h0 = spec_key(h->host.cfg, pol, prop.gcnArchName);
  const int rc = get_spec_module(h->device, h0, kUnitSingle,
                                 [&](std::string& err) { return nsg_spec::spec_compile(h->host.cfg, pol.full, prop.gcnArchName, err, pol.inlane, pol.stream_state); }, &h->spec);
  h->generation++;   // a group that contains this handle re-plans (its specialised unit depends on every member's)
  return rc;
}

int nsg_is_specialized(const nsg_handle* h) { return h && h->spec ? 1 : 0; }
int nsg_spec_origin(const nsg_handle* h) { return h && h->spec ? h->spec->origin : NSG_SPEC_ORIGIN_NONE; }

int nsg_spec_prebuild(const nsg_config* cfg, int64_t n, const char* arch, const char* dir) {
  if (!arch || !*arch || !dir || !*dir) return fail(NSG_EINVAL, "arch and dir are required");
  if (n <= 0 || n > NSG_MAX_ENVS) return fail(NSG_EINVAL, "n must be in [1, 2^27]");
  int rc = validate(cfg, (size_t)kMaxTableBytes);
  if (rc) return rc;
  const SpecPolicy pol = spec_policy(*cfg, n);
  std::string err;
  const std::vector<char> code = nsg_spec::spec_compile(*cfg, pol.full, arch, err, pol.inlane, pol.stream_state);
  if (code.empty()) return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
  return write_unit(dir, spec_key(*cfg, pol, arch), code);
}

int nsg_spec_prebuild_policy(const nsg_config* cfg, int64_t n, int32_t kind, const char* arch, const char* dir) {
  if (!arch || !*arch || !dir || !*dir) return fail(NSG_EINVAL, "arch and dir are required");
  if (kind < 0 || kind > 3) return fail(NSG_EINVAL, "unknown policy kind %d", kind);
  if (n <= 0 || n > NSG_MAX_ENVS) return fail(NSG_EINVAL, "n must be in [1, 2^27]");
  int rc = validate(cfg, (size_t)kMaxTableBytes);
  if (rc) return rc;
  const SpecPolicy pol = spec_policy(*cfg, n);
  std::string err;
  const std::vector<char> code = nsg_spec::policy_compile(*cfg, pol.full, kind, arch, err, pol.inlane, pol.stream_state);
  if (code.empty()) return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
  return write_unit(dir, policy_unit_key(spec_key(*cfg, pol, arch), kind), code);   // the key nsg_rollout_policy looks up
}

int nsg_spec_prebuild_group(const nsg_config* const* cfgs, const int64_t* ns, int32_t count, const char* arch, const char* dir) {
  if (!cfgs || !ns || count <= 0 || count > NSG_MAX_SEGMENTS || !arch || !*arch || !dir || !*dir) return fail(NSG_EINVAL, "bad arguments");
  bool full[NSG_MAX_SEGMENTS];
  uint64_t keys[NSG_MAX_SEGMENTS];
  for (int k = 0; k < count; k++) {
    int rc = validate(cfgs[k], (size_t)kMaxTableBytes);
    if (rc) return rc;
    const SpecPolicy pol = spec_policy(*cfgs[k], ns[k]);
    full[k] = pol.full;
    keys[k] = spec_key(*cfgs[k], pol, arch);
  }
  std::string err;
  const std::vector<char> code = nsg_spec::group_compile(cfgs, full, count, arch, err);
  if (code.empty()) return fail(NSG_EUNSUPPORTED, "%s", err.c_str());
  return write_unit(dir, group_key(keys, count), code);
}

int nsg_destroy(nsg_handle* h) {
  if (!h) return NSG_OK;
  {   // plans this handle is a member of go with it; their tables once nothing can read them any more
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    // their tables are RETIRED, not freed: a destroy may come from a finalizer at any moment - while another thread's group launch
    // is between taking its plan snapshot and enqueueing, or while a stream is capturing (a device synchronisation here would
    // invalidate the capture).  Retired tables are freed by the next drain, which runs on a planning path that has checked for
    // captures and synchronises each table's own device (drain_retired_locked); 16 KB each, at most kMaxRetiredTables parked.
    for (size_t q = 0; q < g_plans.size();) {
      bool member = false;
      for (int k = 0; k < g_plans[q].n_members; k++) member |= g_plans[q].ids[k] == h->id;
      if (member) {
        retire_table_locked(g_plans[q]);
        g_plans.erase(g_plans.begin() + (long)q);
      } else {
        q++;
      }
    }
  }
  if (h->d_tables) (void)hipFree(h->d_tables);
  if (h->d_zig) (void)hipFree(h->d_zig);
  if (h->dev) (void)hipFree(h->dev);
  delete h;
  return NSG_OK;
}

}  // extern "C"
