// igw_device.h -- device-side building blocks of the gridworld step path (gfx950, wave64).
//
// Execution model: GS lanes of a wavefront (GS = 64, 32, ..., 1; a "group") own one env.
// The serial double-precision physics chain is evaluated redundantly by every lane of the
// group (no broadcast needed, the wave issues the instruction anyway); the lanes of a group
// split the ray march of hit_test (by coordinate in 4-lane groups, by sample otherwise).  The
// per-step working set of an env is its 192-byte padded occupancy bitmap, staged in LDS; the int8
// colour grid stays in HBM and is touched only where a colour matters.  Row-sized or rare work
// (histogram updates, resets, Task.__init__) is done by the whole wave, coalesced.
// All arithmetic is IEEE binary64 with one rounding per operation, in the reference's
// operation order (compile with -ffp-contract=off, never fast-math).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/igw.h"
#include "igw_trig.h"
#include "igw_trig_lut.h"

namespace igw {

constexpr int WAVE = 64;
// Issue priority of a wave by its progress through the step (step_kernel only; one hex digit per progress point,
// point 0 = lowest digit).  The SIMD arbitrates issue by priority first and wave age second.  With age alone the
// four co-resident waves of a SIMD run one behind the other and the launch ends with the youngest wave running its
// last phases alone, latency-bound; with "the wave that is behind goes first" all four reach the end together and
// the SIMD stays issue-bound to the end (-9 % launch time, same-box A/B).  `boost`: one level up for a wave that
// knows it has an episode reset to do at the end of the step (-1 %; boosting waves with several changed envs
// measured +1 %).  Points: 0 start, 1 inputs loaded,
// 2 ray march: coordinates done, 3 place / break done, 4 first physics sub-step done, 5 physics done,
// 6 histogram update done, 7 resets done.
#ifndef IGW_PRIO_MAP
#define IGW_PRIO_MAP 0x00011223   /* re-tuned on the round-4 kernel (profiles/r04_ab_variants.txt): round 3's 0x00111233 +0.6 % walking, +1.0 % flying */
#endif
template <bool ON, int PT>
__device__ inline void prio_at(bool boost = false) {
    if constexpr (ON) {
        constexpr int level = (IGW_PRIO_MAP >> (4 * PT)) & 3;
        if (boost) __builtin_amdgcn_s_setprio(level < 3 ? level + 1 : 3);
        else __builtin_amdgcn_s_setprio(level);
    }
}
// Round-6 variants of the step path (same-box A/B: profiles/r06_ab_variants.txt; together -0.5 % walking, -0.4 % CDM):
// the ray march's sample exchange as a 4 x 4 byte transpose of the quad; |p - np| as an operand modifier in the sub-steps
#ifndef IGW_MARCH_TRANSPOSE
#define IGW_MARCH_TRANSPOSE 1
#endif
#ifndef IGW_ABS_SGN
#define IGW_ABS_SGN 1
#endif
#ifndef IGW_BLOCK
#define IGW_BLOCK 256
#endif
constexpr int BLOCK = IGW_BLOCK;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int CELLS = IGW_CELLS;
constexpr int STRIDE = IGW_GRID_STRIDE;
constexpr int CHUNKS = STRIDE / 16;  // 69 dwordx4 per grid row
constexpr int LEVEL = 121;           // cells per y level
// vote histogram: a non-empty target of x-extent wx admits 12 - wx <= 11 shifts dx (same for z), so
// 4 rotations x 11 x 11 bins cover every admissible translation; 16-bit counters, two per word
constexpr int HIST_BINS = 4 * 121;
constexpr int HIST_WORDS = HIST_BINS / 2;  // 242
constexpr int HIST_PAD = 256;
// Occupancy bitmap, the per-step working set of the physics.  HBM: OCC_WORDS dwords per env, one bit per
// cell of the 9 x 13 x 13 box that pads every y level of the build zone with one always-empty cell on each
// side in x and z: bit = (y+1)*169 + (x+6)*13 + (z+6).  LDS: the same words behind a constant prefix that
// holds two empty levels (y <= -3) and the ground plane (y = -2, all ones), followed by a constant empty
// level (y >= 8), so that `key in world` for ANY integer cell is one bit test at
//   idx = L*169 + xp*13 + zp + OCC_IDX0,  L = clamp(y,-4,8)+4, xp = clamp(x,-6,6)+6, zp = clamp(z,-6,6)+6
// with no range checks: clamping sends everything outside the zone to a padding cell or a constant level,
// and all three clamps are [0, 12] after their (even) offsets, i.e. one v_med3_i32 each.
// (The ground plane spans |x|,|z| <= 18; agents stay within |x|,|z| <= 10 -- poses are validated -- and a
// ray is 8 long, so every sample at y = -2 is on it.)
constexpr int OCC_WORDS = IGW_OCC_WORDS;  // 48 dwords = 192 B per env in HBM
constexpr int OCC_LAYER = 169;            // 13 x 13 bits per level
constexpr int OCC_VAR0 = 16;              // LDS word of HBM word 0 (16-byte aligned)
constexpr int OCC_IDX0 = 32 * OCC_VAR0 - 3 * OCC_LAYER;  // 5: level L = 3 (y = -1) starts at LDS word OCC_VAR0
constexpr int OCC_PITCH = 72;             // LDS words per env: 16 constant + 48 variable + 8 constant zero
constexpr int HIST_ROW = IGW_HIST_ROW;    // persistent per-env vote histogram: 512 x u16 (484 used)
static_assert(OCC_IDX0 >= 0 && OCC_IDX0 + 12 * OCC_LAYER + 12 * 13 + 12 < 32 * OCC_PITCH, "LDS occupancy row too short");
static_assert(9 * OCC_LAYER <= 32 * OCC_WORDS, "HBM occupancy row too short");

// gridworld/utils.py:9-24 and core/world.py:9
constexpr double WALKING_SPEED = 5.0;
constexpr double FLYING_SPEED = 15.0;
constexpr double GRAVITY = 20.0;
constexpr double JUMP_SPEED = 0x1.bb67ae8584caap+2;  // sqrt(2 * 20.0 * 1.2) = 6.928203230275509
constexpr double TERMINAL_VELOCITY = 50.0;
constexpr double PAD = 0.25;
constexpr double PI_OVER_180 = 0x1.1df46a2529d39p-6;   // pi / 180  (CPython math.radians)
constexpr double D180_OVER_PI = 0x1.ca5dc1a63c1f8p+5;  // 180 / pi  (CPython math.degrees)

struct alignas(16) AgentRec {  // layout documented in include/igw.h; rewritten whole by every step
    double x, y, z, yaw, pitch, vy;
    int16_t inv[6];    // agent.inventory (20 - blocks of the colour in the world: -1069..20)
    uint16_t step_no;
    uint16_t pack;     // bits 0-1 time_int_steps code (2,4,8,12), bits 2-4 active_block
};
static_assert(sizeof(AgentRec) == IGW_AGENT_BYTES, "agent record layout");

struct alignas(16) AuxRec {  // episode state: read by every step, written only when it changes
    int16_t size;
    uint16_t prev_size;  // bit 15: dirty
    int16_t max_int, target_size;
    int32_t task;
    uint32_t episode;
};
static_assert(sizeof(AuxRec) == IGW_AUX_BYTES, "aux record layout");

struct alignas(16) OutRec {  // per-step outputs (env.py:281-303), written whole by every step
    float agent_pos[5];
    float inventory[6];
    float compass;
    float reward;
    uint8_t done;
    uint8_t pad[11];
};
static_assert(sizeof(OutRec) == IGW_OUT_BYTES && offsetof(OutRec, reward) == 48, "output record layout");

struct alignas(16) TaskMeta {
    double pose[5];
    int16_t target_size, env_max_int;
    uint8_t has_start;
    uint8_t pad0[3];
    int8_t bbox[16];  // offset 48: one aligned dwordx4
    int16_t inv_init[6];
    uint8_t pad[IGW_TASK_META_BYTES - 76];
};
static_assert(offsetof(TaskMeta, bbox) == 48, "bbox must be 16-byte aligned");
static_assert(offsetof(TaskMeta, inv_init) == 64, "inv_init is one aligned dwordx4 (with 4 bytes of padding)");
static_assert(sizeof(TaskMeta) == IGW_TASK_META_BYTES, "task meta layout");

struct KParams {
    int32_t n_envs, select_and_place, size_reward, max_steps, autoreset;
    int32_t debug;  // timing-only ablation switches (igw_config.reserved); 0 in every parity / bench run
    int32_t sample_tasks, n_tasks;  // igw_set_task_sampling: draw env_task uniformly from the table at every reset
    int32_t rt_enabled, rt_max_blocks, rt_levels, rt_max_dist, rt_colors;  // igw_set_random_tasks
    int32_t traj_n, traj_cap;       // igw_set_trajectory_log: envs logged, steps per episode slot
    unsigned long long sample_seed;
    long long env_base;             // igw_config.env_index_base: global index of env 0 (keys the samplers)
    double right_scale, wrong_scale;
    int8_t* grid;
    uint32_t* occ;
    uint16_t* hist;
    AgentRec* agent;
    AuxRec* aux;
    const int8_t* task_target;
    const int8_t* task_start;
    const uint32_t* task_start_occ;
    const TaskMeta* task_meta;
    const uint8_t* task_index;   // [T][IGW_TASK_INDEX_BYTES] colour index of the synthetic targets (include/igw.h)
    OutRec* out;
    unsigned long long* stats;
    uint8_t* traj;               // [traj_n][2][traj_cap][IGW_TRAJ_BYTES]
    int32_t* traj_heads;         // [traj_n][2][4]
    unsigned long long* stamps;  // diagnostic builds only: [waves][8] s_memtime stamps (igw_debug_set_stamps)
};

// ---------------------------------------------------------------- lane groups

// Fixed cross-lane patterns inside a quad (4 consecutive lanes) as DPP moves: one VALU instruction instead
// of a trip through the LDS crossbar (ds_bpermute).  CTRL = quad_perm: bits [2i+1:2i] = source lane of lane i.
template <int CTRL>
__device__ inline int dpp_quad(int v) {  // every quad_perm source lane exists, so there is no `old` value to keep
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ inline double dpp_quad(double v) {
    const int lo = dpp_quad<CTRL>(__double2loint(v)), hi = dpp_quad<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
constexpr int QUAD_BCAST0 = 0x00, QUAD_BCAST1 = 0x55, QUAD_BCAST2 = 0xAA, QUAD_BCAST3 = 0xFF;
constexpr int QUAD_SHIFT_UP = 0x90;  // lane i <- lane i-1, lane 0 keeps its own value
constexpr int QUAD_XOR1 = 0xB1, QUAD_XOR2 = 0x4E;

template <int GS>
struct Grp {
    static constexpr int NG = WAVE / GS;
    int lane;  // lane in wave
    int gl;    // lane in group
    int g;     // group in wave
    __device__ Grp() {
        lane = __lane_id();
        gl = lane & (GS - 1);
        g = lane / GS;
    }
    // ballot restricted to this group, bit i = lane i of the group
    __device__ uint64_t ballot(bool p) const {
        if constexpr (GS == 1) return p ? 1ull : 0ull;
        uint64_t m = __ballot(p);
        if constexpr (GS == 64) return m;
        else return (m >> (g * GS)) & ((1ull << GS) - 1ull);
    }
    // value held by lane `src` of this group (src uniform within the group)
    __device__ int bcast(int v, int src) const {
        if constexpr (GS == 1) return v;
        else if constexpr (GS == 64) return __builtin_amdgcn_readlane(v, src);
        else return __shfl(v, src, GS);
    }
    __device__ int shfl_up1(int v) const {
        if constexpr (GS == 1) return v;
        else if constexpr (GS == 4) return dpp_quad<QUAD_SHIFT_UP>(v);
        else return __shfl_up(v, 1, GS);
    }
    // value of the group's first lane
    __device__ int bcast_first(int v) const {
        if constexpr (GS == 1) return v;
        else if constexpr (GS == 4) return dpp_quad<QUAD_BCAST0>(v);
        else return bcast(v, 0);
    }
    // value of the group's last lane
    __device__ int bcast_last(int v) const {
        if constexpr (GS == 1) return v;
        else if constexpr (GS == 4) return dpp_quad<QUAD_BCAST3>(v);
        else return bcast(v, GS - 1);
    }
    __device__ int group_min(int v) const {
        if constexpr (GS == 4) {
            v = min(v, dpp_quad<QUAD_XOR2>(v));
            return min(v, dpp_quad<QUAD_XOR1>(v));
        } else {
#pragma unroll
            for (int o = GS / 2; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, GS));
            return v;
        }
    }
};

__device__ inline void wave_sync() {
    // LDS traffic of one wave is ordered in hardware; this only stops the compiler
    // from moving LDS accesses of different lanes across the point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// wave-wide maximum of NON-NEGATIVE ints on the DPP network (row shifts, then the two row broadcasts), result
// uniform: no trip through the LDS crossbar.  Lanes without a source read 0.
__device__ inline int wave_max_nonneg(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false));  // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false));  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false));  // row_shr:8  -> lane 15 of a row = row max
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1, 3
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ inline int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- counter RNG
// splitmix64 finaliser; actions = uniform Discrete(18) keyed by (seed, env, t)
__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// uniform task index in [0, n): CustomTasks.reset() (gridworld/tasks/task_set.py:53-56) on the device,
// keyed by (seed, global env index, number of the episode that ends)
__host__ __device__ inline int rng_task(uint64_t seed, uint64_t env, uint64_t episode, int n) {
    uint64_t h = splitmix64(seed ^ splitmix64(env * 0x9E3779B1ull + episode * 0x100000001B3ull + 0x7461736bull));
    return (int)(((h >> 32) * (uint64_t)n) >> 32);
}
// 32-bit counter hash for the random-task generator (murmur3 finaliser over a running combination)
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline uint32_t hash_combine(uint32_t h, uint32_t v) { return fmix32(h ^ (v + 0x9E3779B9u + (h << 6) + (h >> 2))); }
// uniform integer in [0, n) from 32 random bits
__host__ __device__ inline int rng_below(uint32_t r, int n) { return (int)(((uint64_t)r * (uint64_t)n) >> 32); }
__host__ __device__ inline int rng_action18(uint64_t seed, uint64_t env, uint64_t t) {
    uint64_t h = splitmix64(seed ^ splitmix64(env * 0x100000001B3ull + t));
    return (int)(((h >> 32) * 18ull) >> 32);
}

// ---------------------------------------------------------------- env registers

struct Env {  // uniform across the lanes of a group
    double x, y, z, yaw, pitch, vy;
    int step_no, size, prev_size, max_int;
    uint32_t inv01, inv23, inv45;  // agent.inventory: 6 x int16, two per word (the record's layout)
    int tis_code;  // agent.time_int_steps as its code 0..3 = 2, 4, 8, 12 sub-steps (the record's encoding; tis_steps())
    int active, target_size;
    int dirty;  // the histogram changed since max_int was last refreshed (a change with wrong_placement == 0)
    uint32_t episode;  // episodes started (aux record)
};

// inventory[i], i in 0..5 (i is data, not a constant: selects between the three words, BY VALUE -- a select between
// fields of the env struct becomes a load from a selected address and parks the struct in scratch memory)
__device__ __forceinline__ int inv_pick(uint32_t i01, uint32_t i23, uint32_t i45, int i) {
    const uint32_t pair = i < 2 ? i01 : i < 4 ? i23 : i45;
    return (int)(int16_t)(pair >> ((i & 1) << 4));
}
__device__ inline int inv_get(const Env& e, int i) {
    const uint32_t a = e.inv01, b = e.inv23, c = e.inv45;
    return inv_pick(a, b, c, i);
}
// inventory[i] += d as arithmetic on the packed word: the other half of the pair is kept by a bit-field insert, so a carry
// out of the low half (-1 -> 0) cannot reach the high one.  d = 0 leaves everything as it is (callers pass a predicate).
__device__ inline void inv_add(Env& e, int i, int d) {
    const uint32_t a = e.inv01, b = e.inv23, c = e.inv45;
    const int hi = i >> 1;
    const uint32_t sh = (uint32_t)(i & 1) << 4, mask = 0xffffu << sh;
    const uint32_t pair = hi == 0 ? a : hi == 1 ? b : c;
    const uint32_t sum = pair + ((uint32_t)d << sh);
    const uint32_t np = (sum & mask) | (pair & ~mask);
    e.inv01 = hi == 0 ? np : a;
    e.inv23 = hi == 1 ? np : b;
    e.inv45 = hi == 2 ? np : c;
}
__device__ inline int inv_lo(uint32_t pair) { return (int)(int16_t)(pair & 0xffffu); }
__device__ inline int inv_hi(uint32_t pair) { return (int)pair >> 16; }
constexpr uint32_t INV_FULL_PAIR = 20u | (20u << 16);  // Agent.__init__ / reset without a starting grid: 20 of each colour

// time_int_steps of a code (2, 4, 8, 12) and the sub-step length 0.05 / time_int_steps as arithmetic on the code: the
// three powers of two differ in the exponent field only (0.05 / 2 = 0x3f9999999999999a), 0.05 / 12 = 0x3f71111111111111
__device__ inline int tis_steps(int code) { return code == 3 ? 12 : 2 << code; }
__device__ inline double tis_dt(int code) {
    const uint32_t hi = code == 3 ? 0x3f711111u : 0x3f999999u - ((uint32_t)code << 20);
    const uint32_t lo = code == 3 ? 0x11111111u : 0x9999999au;
    return __hiloint2double((int)hi, (int)lo);
}

// The raw words of the two records as the kernels move them: the agent record's fourth 16-byte piece and the aux record.
__device__ inline void env_unpack_piece3(Env& e, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    e.inv01 = w0; e.inv23 = w1; e.inv45 = w2;
    e.step_no = (int)(w3 & 0xffffu);
    e.tis_code = (int)((w3 >> 16) & 3u);
    e.active = (int)((w3 >> 18) & 7u);
}
__device__ inline uint4 env_pack_piece3(const Env& e) {
    return make_uint4(e.inv01, e.inv23, e.inv45, (uint32_t)(e.step_no & 0xffff) | ((uint32_t)e.tis_code << 16) | ((uint32_t)(e.active & 7) << 18));
}
// aux record: {size | prev_size (+ dirty) << 16, max_int | target_size << 16, task, episode}
__device__ inline void env_unpack_aux(Env& e, uint32_t a0, uint32_t a1, uint32_t a3) {
    e.size = (int)(int16_t)(a0 & 0xffffu);
    e.prev_size = (int)((a0 >> 16) & 0x7fffu);
    e.dirty = (int)(a0 >> 31);
    e.max_int = (int)(int16_t)(a1 & 0xffffu);
    e.target_size = (int)a1 >> 16;
    e.episode = a3;
}
__device__ inline uint4 env_pack_aux(const Env& e, int task) {
    return make_uint4(((uint32_t)e.size & 0xffffu) | ((uint32_t)((e.prev_size & 0x7fff) | (e.dirty << 15)) << 16),
                      ((uint32_t)e.max_int & 0xffffu) | ((uint32_t)e.target_size << 16), (uint32_t)task, e.episode);
}
__device__ inline void env_unpack(Env& e, const AgentRec& r, const uint4& aux) {
    e.x = r.x; e.y = r.y; e.z = r.z; e.yaw = r.yaw; e.pitch = r.pitch; e.vy = r.vy;
    uint32_t w[4];
    __builtin_memcpy(w, &r.inv[0], 16);
    env_unpack_piece3(e, w[0], w[1], w[2], w[3]);
    env_unpack_aux(e, aux.x, aux.y, aux.w);
}
// whole-record load (kernels without the spread burst of step_kernel); returns the env's task
__device__ inline int env_load(Env& e, const AgentRec* rec, const AuxRec* aux) {
    const AgentRec r = *rec;  // every lane reads the same 64 B line (one request per wave)
    const uint4 a = *reinterpret_cast<const uint4*>(aux);
    env_unpack(e, r, a);
    return (int)a.z;
}
typedef double vd2 __attribute__((ext_vector_type(2)));
typedef unsigned long long vu2 __attribute__((ext_vector_type(2)));
typedef unsigned int vu4 __attribute__((ext_vector_type(4)));
// Per-step outputs are written once and next read by another launch: non-temporal stores (-1 % launch time).
// (through a pointer in the GLOBAL address space: pointers that went through the kernarg re-read or an inline-asm
// barrier are generic to the compiler, and a flat store also counts on lgkmcnt and may alias LDS for its wait insertion)
#if defined(__HIP_DEVICE_COMPILE__)
#define IGW_GLOBAL(T, p) ((__attribute__((address_space(1))) T*)(uintptr_t)(p))
#else
#define IGW_GLOBAL(T, p) (p)
#endif
template <class T> __device__ inline void st(T* p, T v) { __builtin_nontemporal_store(v, IGW_GLOBAL(T, p)); }
template <class T> __device__ inline void gstore(T* p, T v) { *IGW_GLOBAL(T, p) = v; }   // plain store, global address space
template <class T> __device__ inline T gload(const T* p) { return *IGW_GLOBAL(const T, p); }  // plain load, global address space
__device__ inline void st4(void* p, const uint4& v) { st(reinterpret_cast<vu4*>(p), vu4{v.x, v.y, v.z, v.w}); }
__device__ inline void env_store_pose(const Env& e, AgentRec* rec) {
    vd2* d = reinterpret_cast<vd2*>(rec);
    st(d + 0, vd2{e.x, e.y});
    st(d + 1, vd2{e.z, e.yaw});
    st(d + 2, vd2{e.pitch, e.vy});
}
// one lane stores the whole agent record (and, separately, the aux record)
__device__ inline void env_store(const Env& e, AgentRec* rec) {
    env_store_pose(e, rec);
    st4(reinterpret_cast<uint4*>(rec) + 3, env_pack_piece3(e));
}
__device__ inline void aux_store(const Env& e, int task, AuxRec* aux) { st4(aux, env_pack_aux(e, task)); }

// The output record (include/igw.h) as its four 16-byte pieces: agentPos[0..3] | agentPos[4], inventory[0..2] |
// inventory[3..5], compass | reward, done.
__device__ inline uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ inline uint4 out_piece(const Env& e, int q, float a0, float a1, float a2, float a3, float a4, float compass) {
    const float i0 = (float)inv_lo(e.inv01), i1 = (float)inv_hi(e.inv01), i2 = (float)inv_lo(e.inv23),
                i3 = (float)inv_hi(e.inv23), i4 = (float)inv_lo(e.inv45), i5 = (float)inv_hi(e.inv45);
    return make_uint4(f2u(q == 0 ? a0 : q == 1 ? a4 : i3), f2u(q == 0 ? a1 : q == 1 ? i0 : i4),
                      f2u(q == 0 ? a2 : q == 1 ? i1 : i5), f2u(q == 0 ? a3 : q == 1 ? i2 : compass));
}
__device__ inline uint4 out_piece_step(const Env& e, int q) {  // obs of step(): env.py:281-289
    return out_piece(e, q, (float)e.x, (float)e.y, (float)e.z, (float)e.pitch, (float)e.yaw, (float)(e.yaw - 180.0));
}
__device__ inline uint4 out_piece_reset(const Env& e, int q) {  // obs of reset(): agentPos zeros, compass 0 (env.py:247-254)
    return out_piece(e, q, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
}
__device__ inline uint4 out_piece_result(float reward, bool done) { return make_uint4(f2u(reward), done ? 1u : 0u, 0u, 0u); }
// one lane writes the whole record
__device__ inline void out_store(OutRec* o, const Env& e, bool reset_obs, float reward, bool done) {
#pragma unroll
    for (int q = 0; q < 3; q++) st4(reinterpret_cast<uint4*>(o) + q, reset_obs ? out_piece_reset(e, q) : out_piece_step(e, q));
    st4(reinterpret_cast<uint4*>(o) + 3, out_piece_result(reward, done));
}

// A register holding ANY value, at no cost (LLVM: freeze poison).  For the lanes of a divergent branch that do not take it
// when the others load something: merged with a zero, the loaded value would be needed (copied) where the branch ends --
// and the wavefront would wait for the load there; merged with "any value" nothing is needed before the first real use.
// (Not `int x;` read uninitialised: that is undefined behaviour, this is not.)
__device__ inline uint32_t any_value_u() {
    uint32_t v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    v = __builtin_nondeterministic_value(v);
#endif
    return v;
}
__device__ inline int any_value() { return (int)any_value_u(); }

// ---------------------------------------------------------------- world queries

// core/world.py:57-58
__device__ inline bool build_zone_d(double x, double y, double z, double pad) {
    // -5-pad <= x <= 5+pad  <=>  |x| <= 5+pad (exact); likewise z
    return (__builtin_fabs(x) <= 5.0 + pad) & (__builtin_fabs(z) <= 5.0 + pad) & (-1.0 - pad <= y) & (y < 8.0 + pad);
}
__device__ inline bool build_zone_i(int x, int y, int z) {
    return ((unsigned)(x + 5) <= 10u) & ((unsigned)(z + 5) <= 10u) & ((unsigned)(y + 1) <= 8u);
}
__device__ inline int cell_of(int x, int y, int z) { return (y + 1) * LEVEL + (x + 5) * 11 + (z + 5); }

// bit of a build-zone cell in the HBM occupancy row / of any integer cell in the LDS row
__device__ inline int occ_bit_hbm(int x, int y, int z) { return (y + 1) * OCC_LAYER + (x + 6) * 13 + (z + 6); }
__device__ inline int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }  // one v_med3_i32
__device__ inline int occ_idx(int x, int y, int z) {
    return (clampi(y, -4, 8) + 4) * OCC_LAYER + (clampi(x, -6, 6) + 6) * 13 + (clampi(z, -6, 6) + 6) + OCC_IDX0;
}
// occ_idx of the three clamped, offset coordinates packed in one word (xp | L << 8 | zp << 16): one v_dot4_u32_u8
__device__ inline int key_idx(int key) {
    return (int)__builtin_amdgcn_udot4((unsigned)key, 0x0001A90Du, (unsigned)OCC_IDX0, false);
}
__device__ inline bool occ_test(const uint32_t* occ_s, int idx) { return (occ_s[idx >> 5] >> (idx & 31)) & 1u; }

// `key in world` (World.world dict membership, core/world.py:60-71 ground plane + placed blocks): one LDS
// bit test, no control flow.  Colours live in the int8 grid row in HBM and are only fetched for the one
// block a break hits.
__device__ inline bool world_has(const uint32_t* occ_s, int x, int y, int z) { return occ_test(occ_s, occ_idx(x, y, z)); }

// constant words of one env's LDS occupancy row (see OCC_IDX0): words 0..15 = two empty levels + the ground
// plane, words 64..71 = zero (the empty level above the zone reaches word 68)
__host__ __device__ constexpr uint32_t occ_const_word(int w) {
    constexpr int g0 = OCC_IDX0 + 2 * OCC_LAYER, g1 = OCC_IDX0 + 3 * OCC_LAYER;  // ground plane bits [g0, g1)
    if (w >= OCC_VAR0) return 0u;
    const int lo = g0 - 32 * w > 0 ? g0 - 32 * w : 0, hi = g1 - 32 * w < 32 ? g1 - 32 * w : 32;
    if (hi <= lo) return 0u;
    return (hi - lo == 32) ? 0xffffffffu : (((1u << (hi - lo)) - 1u) << lo);
}

// ---------------------------------------------------------------- trig front-end

struct TrigCtx {
    const double* lut;  // IGW_TRIG_LUT in constant memory: {cos, sin} of radians(5k), k = IGW_LUT_K0..
};

// cos / sin of math.radians(deg).  Multiples of 5 degrees inside the table (all that discrete
// walking can produce) come from the LUT; anything else from the build's own
// double-precision sincos (igw_trig.h).
// LUT = false (flying: angles are arbitrary floats, practically never on the lattice) goes straight to the general
// evaluation, which returns the same bits on the lattice (tests/test_trig.py).
template <bool LUT = true>
__device__ inline void sincos_deg(const TrigCtx& t, double deg, double& s, double& c) {
    if constexpr (!LUT) {
        igw_sincos(deg * PI_OVER_180, &s, &c);
        return;
    }
    const int id = (int)deg;
    // on the 5-degree lattice inside the table?  u = id - first angle; k = u / 5 by a 24-bit multiply (u <= 630 < 2^10:
    // u * 0xCCCD >> 18 is exact there; v_mul_u32_u24 is full rate, the 32-bit multiply of a general id % 5 quarter rate)
    const unsigned u = (unsigned)(id - 5 * IGW_LUT_K0);
    const unsigned k = __umul24(u & 0x3ffu, 0xCCCDu) >> 18;
    if ((double)id == deg && u <= 5u * (IGW_LUT_N - 1) && __umul24(k, 5u) == u) {
        c = t.lut[2 * k];
        s = t.lut[2 * k + 1];
    } else {
        igw_sincos(deg * PI_OVER_180, &s, &c);
    }
}

// normalize(): int(round(x)), round-half-even (gridworld/utils.py:57-73), as ONE f64 add: adding
// 1.5 * 2^52 leaves the integer nearest to x (ties to even, default rounding mode) in the low mantissa
// bits, i.e. in the low dword in two's complement.  Exact for |x| < 2^31; positions here stay within +-40.
__device__ inline int rint_i32(double x) {
    return __double2loint(x + 0x1.8p52);
}

// x / 5 exactly as IEEE division rounds it, in 3 flops instead of a full f64 division sequence:
// q0 = RN(x * RN(1/5)), r = x - 5 q0 (exact in an fma), q = RN(q0 + r * RN(1/5)) is the correctly rounded
// quotient (Markstein's theorem; checked bit-for-bit against x / 5.0 on 4*10^8 arguments).  Values near the
// underflow range (where the residual may be inexact) take the real division.
__device__ inline double div5(double x) {
    // zero takes the fast path too: it yields a zero, and the sign of a zero step cannot change any sample
    // (p + -0 == p + +0 for every p but -0, and normalize(+-0) == 0 either way)
    if (__builtin_fabs(x) >= 0x1p-900 || x == 0.0) {
        const double c = 0x1.999999999999ap-3;
        const double q = x * c;
        const double r = __builtin_fma(-5.0, q, x);
        return __builtin_fma(r, c, q);
    }
    return x / 5.0;
}

// ---------------------------------------------------------------- collide (core/world.py:264-310)

// Neighbourhood prober for collide: the 12 probes are np + small offsets; each is one clamped index and
// one LDS bit test (occ_idx).
struct Probe {
    const uint32_t* occ_s;
    int nx, ny, nz;
    __device__ bool at(int dx, int dy, int dz) const { return world_has(occ_s, nx + dx, ny + dy, nz + dz); }
};

// Six faces in the reference order; a face only probes when its overlap test passes.  (Issuing all 12
// probes up front -- they depend only on np -- was measured slower at every group size.)
__device__ inline void collide(Env& e, const uint32_t* occ_s, double& px, double& py, double& pz) {
    const int nx = rint_i32(px), ny = rint_i32(py), nz = rint_i32(pz);
    const Probe w{occ_s, nx, ny, nz};
    double d;
    d = (py - (double)ny) * 1.0;  // face (0, 1, 0): heights dy = 0, 1 probe (nx, ny - dy + 1, nz)
    if (!(d < PAD)) {
        if (w.at(0, 1, 0) || w.at(0, 0, 0)) {
            py -= (d - PAD) * 1.0;
            e.vy = 0.0;
        }
    }
    d = (py - (double)ny) * -1.0;  // face (0, -1, 0)
    if (!(d < PAD)) {
        if (w.at(0, -1, 0) || w.at(0, -2, 0)) {
            py -= (d - PAD) * -1.0;
            e.vy = 0.0;
        }
    }
    d = (px - (double)nx) * -1.0;  // face (-1, 0, 0)
    if (!(d < PAD)) {
        if (w.at(-1, 0, 0) || w.at(-1, -1, 0)) px -= (d - PAD) * -1.0;
    }
    d = (px - (double)nx) * 1.0;  // face (1, 0, 0)
    if (!(d < PAD)) {
        if (w.at(1, 0, 0) || w.at(1, -1, 0)) px -= (d - PAD) * 1.0;
    }
    d = (pz - (double)nz) * 1.0;  // face (0, 0, 1)
    if (!(d < PAD)) {
        if (w.at(0, 0, 1) || w.at(0, -1, 1)) pz -= (d - PAD) * 1.0;
    }
    d = (pz - (double)nz) * -1.0;  // face (0, 0, -1)
    if (!(d < PAD)) {
        if (w.at(0, 0, -1) || w.at(0, -1, -1)) pz -= (d - PAD) * -1.0;
    }
}

// collide with the three axes spread over the lanes of a group (groups of 4+ lanes).  The reference visits
// the faces in the order y+, y-, x-, x+, z+, z-; a face reads and writes only its own coordinate and its
// probes depend only on np, so the axes are independent and only the two faces of one axis are ordered.
// Lane (gl & 3) = 0 / 3 takes y, 1 takes x, 2 takes z, with exactly the reference's arithmetic; the three
// results (and dy, which only the y faces clear) are then exchanged inside the group.  Everything is
// predicated, not branched: the lanes of a wave disagree on almost every test.
template <int GS>
__device__ inline void collide_split(const Grp<GS>& G, Env& e, const uint32_t* occ_s, double& px, double& py,
                                     double& pz) {
    static_assert(GS >= 4, "needs three lanes per env");
    const int nx = rint_i32(px), ny = rint_i32(py), nz = rint_i32(pz);
    const int a = G.gl & 3;
    const bool ax = a == 1, az = a == 2;
    const int ux = ax ? 1 : 0, uz = az ? 1 : 0, uy = (ax || az) ? 0 : 1;
    double pa = ax ? px : az ? pz : py;
    const double na = (double)(ax ? nx : az ? nz : ny);
    double vy = e.vy;
    // The reference tries both faces of an axis in turn, (0,1,0) before (0,-1,0), (-1,0,0) before (1,0,0), (0,0,1)
    // before (0,0,-1), each with d = (p - np) * f and `d < PAD => skip`.  |p - np| <= 0.5, so d >= PAD can hold for
    // ONE of the two faces only -- the one on the side of the cell centre the agent is on -- and after its push
    // d is exactly -PAD for the other: only that face is probed (two cells, one level apart), same arithmetic.
    const double sgn = pa - na;
    const bool pos = sgn > 0.0;                  // the face with f = +1 along this axis, else f = -1
    const int i1 = pos ? 1 : -1;
    const double f1 = pos ? 1.0 : -1.0;
    // the two probe cells differ from np only along this lane's axis (and one level down)
    const int xa = (clampi(nx + ux * i1, -6, 6) + 6) * 13;
    const int za = clampi(nz + uz * i1, -6, 6) + 6 + OCC_IDX0;
    const int ya0 = (clampi(ny + uy * i1, -4, 8) + 4) * OCC_LAYER, ya1 = (clampi(ny + uy * i1 - 1, -4, 8) + 4) * OCC_LAYER;
    const bool b1 = (int)occ_test(occ_s, ya0 + xa + za) | (int)occ_test(occ_s, ya1 + xa + za);
    const double d = sgn * f1;
    const bool h1 = !(d < PAD) && b1;
    pa = h1 ? pa - (d - PAD) * f1 : pa;
    if (uy && h1) vy = 0.0;
    py = dpp_quad<QUAD_BCAST0>(pa);
    px = dpp_quad<QUAD_BCAST1>(pa);
    pz = dpp_quad<QUAD_BCAST2>(pa);
    e.vy = dpp_quad<QUAD_BCAST0>(vy);
}

// The sub-steps of a step (world_update) for groups of four or more lanes, axis-owned: lane (gl & 3) = 1 / 2 / 0, 3 of a
// group carries ONLY the x / z / y coordinate through all the sub-steps -- its own displacement, its own candidate
// position, its own collision face (collide_split's arithmetic, unchanged) -- and the lanes exchange just the rounded
// cell coordinates (three one-dword broadcasts per sub-step) for the probe index, plus, when the wavefront is not known
// to stay inside the padded zone (!INSIDE: flying always; walkers near the border), one bit per axis for the zone test.
// collide_split handed every lane all three positions after every sub-step: six broadcast dwords, three candidate
// positions and three roundings per lane, and the selects that pick the lane's own out of them again.
// The x / z lanes run the y lanes' velocity code on a neutral element: their "velocity" is -0.0 and their "gravity"
// +0.0, so vy stays -0.0 (-0.0 - 0.0, max(-0.0, -50)), vy * dt is -0.0, and displacement + -0.0 is the displacement
// bit for bit (x + -0.0 == x for every x, -0.0 included: no select needed to keep the axes apart).
// Returns vy as it was before the last sub-step's clamp (time_int_steps, core/world.py:243-250).
template <int GS, bool FLY, bool INSIDE>
__device__ inline double substeps_owned(const Grp<GS>& G, Env& e, const uint32_t* occ_s, double mvx, double mvy, double mvz,
                                        int m, double dt) {
    static_assert(GS >= 4, "needs three lanes per env");
    const int a = G.gl & 3;
    const bool ax = a == 1, az = a == 2, ay = !(ax || az);
    const int ux = ax ? 1 : 0, uz = az ? 1 : 0, uy = ay ? 1 : 0;
    double pa = ax ? e.x : az ? e.z : e.y;
    const double ca = (ax ? mvx : az ? mvz : mvy) * (dt * (FLY ? FLYING_SPEED : WALKING_SPEED));   // ddx / ddz / the motion part of ddy
    const double ga = ay ? dt * GRAVITY : 0.0;
    double vy = ay ? e.vy : -0.0, vy_pre = 0.0;
    for (int i = 0; i < m; i++) {  // _update, core/world.py:222-262
        if constexpr (!FLY) {
            vy -= ga;
            vy_pre = vy;
            vy = vy > -TERMINAL_VELOCITY ? vy : -TERMINAL_VELOCITY;
        }
        double c = pa + (ca + vy * dt);
        bool moves = true;   // flying outside the padded zone: the sub-step does nothing at all
        if constexpr (!INSIDE) {   // build_zone(candidate, pad = 2), one comparison pair per axis lane
            const int ok = ay ? (int)((-3.0 <= c) & (c < 10.0)) : (int)(__builtin_fabs(c) <= 7.0);
            const bool zone = (dpp_quad<QUAD_BCAST0>(ok) & dpp_quad<QUAD_BCAST1>(ok) & dpp_quad<QUAD_BCAST2>(ok)) != 0;
            if constexpr (FLY) moves = zone;
            else c = (zone || ay) ? c : pa;   // outside it a walker only moves vertically
        }
        const int n = rint_i32(c);
        const int ny = dpp_quad<QUAD_BCAST0>(n), nx = dpp_quad<QUAD_BCAST1>(n), nz = dpp_quad<QUAD_BCAST2>(n);
        // (from here on: collide_split, with this lane's axis only)
        const double sgn = c - (double)n;
        const bool pos = sgn > 0.0;
        const int i1 = pos ? 1 : -1;
        const double f1 = pos ? 1.0 : -1.0;
        const int xa = (clampi(nx + ux * i1, -6, 6) + 6) * 13;
        const int za = clampi(nz + uz * i1, -6, 6) + 6 + OCC_IDX0;
        const int ya0 = (clampi(ny + uy * i1, -4, 8) + 4) * OCC_LAYER, ya1 = (clampi(ny + uy * i1 - 1, -4, 8) + 4) * OCC_LAYER;
        const bool b1 = (int)occ_test(occ_s, ya0 + xa + za) | (int)occ_test(occ_s, ya1 + xa + za);
#if IGW_ABS_SGN
        const double d = __builtin_fabs(sgn);   // == sgn * f1 exactly (f1 = +-1 with the sign of sgn): an operand modifier, no instruction
#else
        const double d = sgn * f1;
#endif
        const bool h1 = !(d < PAD) && b1;
        const double r = h1 ? c - (d - PAD) * f1 : c;
        pa = moves ? r : pa;
        if (uy && h1 && moves) vy = 0.0;
    }
    e.y = dpp_quad<QUAD_BCAST0>(pa);
    e.x = dpp_quad<QUAD_BCAST1>(pa);
    e.z = dpp_quad<QUAD_BCAST2>(pa);
    e.vy = dpp_quad<QUAD_BCAST0>(vy);
    return dpp_quad<QUAD_BCAST0>(vy_pre);
}

// ---------------------------------------------------------------- hit_test (core/world.py:73-99)

struct Hit {
    bool hit, have_prev;
    int bx, by, bz;  // the block that was hit (ground when by == -2); outside the zone: clamped (see occ_idx)
    int px, py, pz;  // `previous`: the last empty cell in front of it (clamped likewise)
};

// A sample's cell as the three clamped, offset coordinates of occ_idx packed in one word:
//   key = xp | L << 8 | zp << 16,  xp = clamp(x,-6,6)+6, L = clamp(y,-4,8)+4, zp = clamp(z,-6,6)+6.
// The offsets are EVEN so they can ride in the rounding constant: x + (1.5 * 2^52 + 6) rounds to the same
// integer (ties to even keep their parity) and leaves normalize(x) + 6 in the low dword (rint_i32).
// Clamped keys answer everything hit_test is asked: cells outside the zone are only ever "in world" on the
// ground plane, whose first sample always differs from its predecessor in L; `previous` matters only inside
// the zone, where nothing is clamped (the clamp range strictly contains the zone).
constexpr double RINT_MAGIC = 0x1.8p52;
__device__ inline int key_of(double x, double y, double z) {
    const int xp = clampi(__double2loint(x + (RINT_MAGIC + 6.0)), 0, 12);
    const int l1 = clampi(__double2loint(y + (RINT_MAGIC + 4.0)), 0, 12);
    const int zp = clampi(__double2loint(z + (RINT_MAGIC + 6.0)), 0, 12);
    return xp | (l1 << 8) | (zp << 16);
}
__device__ inline void key_unpack(int key, int& x, int& y, int& z) {
    x = (key & 0xff) - 6; y = ((key >> 8) & 0xff) - 4; z = ((key >> 16) & 0xff) - 6;
}

// byte `j` of `w` := min(v, lim) (unsigned), the other bytes of w untouched (j == 0: zeroed): v_min_u32 with a byte
// destination.
__device__ inline void sdwa_min_into_byte(uint32_t& w, uint32_t v, uint32_t lim, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (j == 0) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(w) : "v"(v), "v"(lim));  // (byte 0 starts a new word: the other bytes := 0)
    else if (j == 1) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(v), "v"(lim));
    else if (j == 2) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(v), "v"(lim));
    else asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(v), "v"(lim));
#else
    w = (j == 0 ? 0u : (w & ~(0xffu << (8 * j)))) | ((v < lim ? v : lim) << (8 * j));
#endif
}

// Sample k is position + k sequential additions of vector/5 (the reference's recurrence, so the
// rounding of every partial sum is reproduced).
// `scratch`: WAVE * 10 words of LDS owned by the wave (groups of four lanes only; see there).
template <int GS, bool PRIO = false>
__device__ inline Hit hit_test(const Grp<GS>& G, const uint32_t* occ_s, double x, double y, double z,
                               double vx, double vy, double vz, bool boost = false, uint32_t* scratch = nullptr) {
    constexpr int SAMPLES = 40;  // max_distance 8 * m 5
    Hit h;
    h.hit = false; h.have_prev = false;
    h.bx = h.by = h.bz = 0;
    h.px = h.py = h.pz = 0;
    if constexpr (GS == 4) {
        // Coordinate split: lane 0 / 1 / 2 of the group carries the x / y / z recurrence alone (39 sequential
        // adds instead of 117 per lane; one division by 5 instead of three) and rounds + clamps its coordinate of
        // all 40 samples; four samples at a time travel as the four bytes of one word per coordinate, are broadcast
        // inside the quad, and lane j assembles the key of sample 4r + j from byte j of the three words, probes the
        // occupancy row and parks the key in the wave's LDS scratch.
        //
        // The reference returns the first sample with `key != previous and key in world`.  Membership is a function
        // of the key alone and the world does not change during the march, so the FIRST SAMPLE IN THE WORLD always
        // differs from its predecessor (which was not in the world): the hit is the earliest set membership bit.
        // Each lane folds its ten bits into a mask; one group minimum of 4 * round + lane finds the sample; its
        // key and the one before it (`previous`) are then read back from the scratch.
        constexpr int ROUNDS = SAMPLES / 4;
        const int a = G.gl & 3;
        double c = a == 0 ? x : a == 1 ? y : z;
        const double sc = div5(a == 0 ? vx : a == 1 ? vy : vz);  // dx / m with m = 5
        const double magic = RINT_MAGIC + (a == 1 ? 4.0 : 6.0);
        // byte selectors (v_perm_b32: 0-3 = bytes of the second source, 4-7 = bytes of the first, 0x0c = zero)
#if IGW_MARCH_TRANSPOSE
        // the 4 x 4 byte transpose of a quad in two exchange stages (lane ^ 1, lane ^ 2): one v_perm_b32 per stage
        const unsigned sel_1 = (G.gl & 1) ? 0x03070105u : 0x06020400u;   // odd: (partner1, own1, partner3, own3); even: (own0, partner0, own2, partner2)
        const unsigned sel_2 = (G.gl & 2) ? 0x03020706u : 0x05040100u;   // lanes 2, 3: (partner2, partner3, own2, own3); lanes 0, 1: (own0, own1, partner0, partner1)
#else
        const unsigned sel_xy = 0x0c0c0000u | (unsigned)G.gl | ((4u + (unsigned)G.gl) << 8);
        const unsigned sel_z = 0x0c000100u | ((4u + (unsigned)G.gl) << 16);
#endif
        uint32_t* const mine = scratch + G.lane;
        int kidx[ROUNDS];
        uint32_t word[ROUNDS];
        const uint32_t twelve = 12u;
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            uint32_t w;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // byte j of w = min(unsigned(rounded + offset), 12) in ONE instruction (a sub-dword destination,
                // SDWA).  The UNSIGNED minimum sends negative values to 12 where the signed clamp gave 0; both
                // name cells that are never occupied (a padding column, an empty level outside the zone) and never
                // in the build zone, and that is all hit_test and place / break ask of a clamped coordinate.
                sdwa_min_into_byte(w, (uint32_t)__double2loint(c + magic), twelve, j);
                if (r * 4 + j + 1 < SAMPLES) c = c + sc;
            }
#if IGW_MARCH_TRANSPOSE
            // Lane c of the quad holds the four samples of coordinate c (x, y, z, z) as the bytes of w; lane j needs byte j of
            // every lane: a 4 x 4 byte transpose, two DPP exchanges + two v_perm_b32 (three broadcasts + two perms before).
            // Byte 3 of the key is lane 3's copy of z: key_idx() weighs it with zero, key_unpack() masks it off.
            const unsigned u1 = __builtin_amdgcn_perm((unsigned)dpp_quad<QUAD_XOR1>((int)w), w, sel_1);
            const int key = (int)__builtin_amdgcn_perm((unsigned)dpp_quad<QUAD_XOR2>((int)u1), u1, sel_2);
#else
            const unsigned wx = (unsigned)dpp_quad<QUAD_BCAST0>((int)w), wy = (unsigned)dpp_quad<QUAD_BCAST1>((int)w),
                           wz = (unsigned)dpp_quad<QUAD_BCAST2>((int)w);
            // byte gl of wx, wy, wz -> bytes 0, 1, 2 of the key
            const int key = (int)__builtin_amdgcn_perm(wz, __builtin_amdgcn_perm(wy, wx, sel_xy), sel_z);
#endif
            kidx[r] = key_idx(key);
            word[r] = occ_s[kidx[r] >> 5];  // all probes are issued before the first is consumed
            mine[r * WAVE] = (uint32_t)key;
        }
        prio_at<PRIO, 2>(boost);
        uint32_t m = 0;
#pragma unroll
        for (int r = ROUNDS - 1; r >= 0; r--)
            m = (m << 1) | __builtin_amdgcn_ubfe(word[r], (uint32_t)kidx[r], 1u);  // bit r: sample 4r + gl is in the world
        const int first = G.group_min(m ? (__builtin_ctz(m) << 2) | a : SAMPLES);
        wave_sync();
        if (first < SAMPLES) {
            uint32_t* const quad = scratch + (G.lane & ~3);
            const int before = first > 0 ? first - 1 : 0;
            const int bk = (int)quad[(first >> 2) * WAVE + (first & 3)];
            const int pk = (int)quad[(before >> 2) * WAVE + (before & 3)];
            h.hit = true;
            h.have_prev = first != 0;
            key_unpack(bk, h.bx, h.by, h.bz);
            key_unpack(pk, h.px, h.py, h.pz);
        }
        return h;
    }
    const double sx = div5(vx), sy = div5(vy), sz = div5(vz);  // dx / m with m = 5
    if constexpr (GS == 1) {
        // one lane per env: the reference loop as is; stop when every active lane has its answer
        int q = 0, bk = 0, pk = 0;
        for (int s = 0; s < SAMPLES; s++) {
            const int k = key_of(x, y, z);
            if (!h.hit && ((s == 0) || k != q) && occ_test(occ_s, key_idx(k))) {
                h.hit = true;
                h.have_prev = s != 0;
                bk = k; pk = q;
            }
            if (!__any(!h.hit)) break;
            q = k;
            x = x + sx; y = y + sy; z = z + sz;
        }
        key_unpack(bk, h.bx, h.by, h.bz);
        key_unpack(pk, h.px, h.py, h.pz);
        return h;
    } else {
        // Sample split.  Groups of 2: lane j owns the contiguous samples j*C .. j*C+C-1 (it first walks to its
        // chunk -- the reference's recurrence, nothing can be skipped).  Wider groups: lane j owns samples
        // j, j+GS, ... (every round after the first advances all lanes by GS unmasked adds).
        constexpr bool CHUNKED = GS == 2;
        constexpr int C = (SAMPLES + GS - 1) / GS;  // samples per lane (chunked) = rounds (strided)
        int key[C];
        bool inw[C];
        if constexpr (CHUNKED) {
            const int lead = G.gl * C;
            for (int i = 0; i < (GS - 1) * C && i < SAMPLES - 1; i++) {
                if (i < lead) { x = x + sx; y = y + sy; z = z + sz; }
            }
#pragma unroll
            for (int k = 0; k < C; k++) {
                key[k] = key_of(x, y, z);
                inw[k] = occ_test(occ_s, key_idx(key[k]));
                if (k + 1 < C) { x = x + sx; y = y + sy; z = z + sz; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < C; r++) {
                int nadd = (r == 0) ? G.gl : GS;
                if (GS == 64) nadd = min(nadd, SAMPLES - 1);
                const int bound = (r == 0) ? min(GS - 1, SAMPLES - 1) : GS;
                if (r == 0) {
                    for (int i = 0; i < bound; i++) {
                        if (i < nadd) { x = x + sx; y = y + sy; z = z + sz; }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < bound; i++) { x = x + sx; y = y + sy; z = z + sz; }
                }
                key[r] = key_of(x, y, z);
                inw[r] = occ_test(occ_s, key_idx(key[r]));
            }
        }
        // The hit is the first sample in the world (it always differs from its predecessor, see the four-lane path):
        // every lane finds its earliest one, a group minimum names the sample, and its key and the key of the sample
        // before it are fetched from their owners.
        int first = SAMPLES, fkey = 0;
#pragma unroll
        for (int k = C - 1; k >= 0; k--) {  // descending, so the earliest wins
            const int s = CHUNKED ? G.gl * C + k : k * GS + G.gl;
            if (s < SAMPLES && inw[k]) { first = s; fkey = key[k]; }
        }
        const int best = G.group_min(first);
        if (best < SAMPLES) {
            const int before = best > 0 ? best - 1 : 0;
            const int owner = CHUNKED ? best / C : best % GS, owner_b = CHUNKED ? before / C : before % GS;
            const int slot_b = CHUNKED ? before % C : before / GS;  // where the owner of `before` keeps it
            int bkey = key[0];
#pragma unroll
            for (int k = 1; k < C; k++) bkey = (k == slot_b) ? key[k] : bkey;
            const int bk = G.bcast(fkey, owner), pk = G.bcast(bkey, owner_b);
            h.hit = true;
            h.have_prev = best != 0;
            key_unpack(bk, h.bx, h.by, h.bz);
            key_unpack(pk, h.px, h.py, h.pz);
        }
        return h;
    }
}

// ---------------------------------------------------------------- maximal_intersection, full vote
// Used where there is no running histogram: Task.__init__ (GridWorld.max_int of the user task on the
// starting grid) and the stateless igw_task_eval.  The step kernels update a persistent histogram
// incrementally instead (resolve_changes in igw_kernels.hip).
// (tasks/task.py:147-161 restated as a vote: every (target block, grid block) pair on the same
// level with equal non-zero value votes for translation (tx - gx, tz - gz) of each rotation; the
// answer is the best admissible bin.  Admissible set == bounding-box rule, tasks/task.py:62-72.)

struct MiResult {
    int max_int;
    int arg_dx, arg_dz, arg_rot;
};

// Whole wave, one env.  grid_s: LDS copy of the env's int8 grid row; start_g: global starting grid row or nullptr
// (grid values are compared as grid - start); tgt_s: LDS copy of the target row (rotation 0);
// hist: HIST_PAD words of LDS; bbox: 4 x (xmin, xmax, zmin, zmax) packed one int per rotation.
template <bool ARGMAX>
__device__ inline MiResult max_intersection_wave(const int8_t* grid_s, const int8_t* start_g,
                                                 const int8_t* tgt_s, uint32_t* hist, const int* bbox4,
                                                 int nrot) {
    const int lane = __lane_id();
    MiResult res;
    res.max_int = 0;
    res.arg_dx = res.arg_dz = res.arg_rot = 0;
    // empty target (bbox xmin > xmax): every translation is admissible but nothing can match
    if ((int8_t)(bbox4[0] & 0xff) > (int8_t)((bbox4[0] >> 8) & 0xff)) return res;
    for (int w = lane; w < HIST_PAD; w += WAVE) hist[w] = 0;
    wave_sync();
    const int c0 = lane, c1 = lane + 64;
    const bool v1 = c1 < LEVEL;
    const int tx0 = c0 / 11, tz0 = c0 % 11;
    const int tx1 = v1 ? c1 / 11 : 0, tz1 = v1 ? c1 % 11 : 0;
    int bb[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        bb[r][0] = (int8_t)(bbox4[r] & 0xff);
        bb[r][1] = (int8_t)((bbox4[r] >> 8) & 0xff);
        bb[r][2] = (int8_t)((bbox4[r] >> 16) & 0xff);
        bb[r][3] = (int8_t)((bbox4[r] >> 24) & 0xff);
    }
    auto vote = [&](int tx, int tz, int gx, int gz) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (r < nrot) {
                // rotation r of target cell (x, z): (x,z) -> (z, 10-x) -> (10-x, 10-z) -> (10-z, x)
                const int rx = r == 0 ? tx : r == 1 ? tz : r == 2 ? 10 - tx : 10 - tz;
                const int rz = r == 0 ? tz : r == 1 ? 10 - tx : r == 2 ? 10 - tz : tx;
                const int dx = rx - gx, dz = rz - gz;
                const int dxlo = bb[r][1] - 10, dzlo = bb[r][3] - 10;
                if (dx >= dxlo && dx <= bb[r][0] && dz >= dzlo && dz <= bb[r][2]) {
                    const int bin = r * 121 + (dx - dxlo) * 11 + (dz - dzlo);
                    atomicAdd(&hist[bin >> 1], 1u << (16 * (bin & 1)));
                }
            }
        }
    };
    for (int y = 0; y < IGW_GRID_Y; y++) {
        const int base = y * LEVEL;
        int g0 = grid_s[base + c0];
        int g1 = v1 ? grid_s[base + c1] : 0;
        if (start_g) {
            g0 -= start_g[base + c0];
            if (v1) g1 -= start_g[base + c1];
        }
        const int t0 = tgt_s[base + c0];
        const int t1 = v1 ? tgt_s[base + c1] : 0;
        uint64_t gm0 = __ballot(g0 != 0), gm1 = __ballot(g1 != 0);
        const uint64_t tany = __ballot((t0 | t1) != 0);
        if ((gm0 | gm1) == 0 || tany == 0) continue;
        while (gm0) {
            const int b = __builtin_ctzll(gm0);
            gm0 &= gm0 - 1;
            const int val = __builtin_amdgcn_readlane(g0, b);
            const int gx = b / 11, gz = b % 11;
            if (t0 == val) vote(tx0, tz0, gx, gz);
            if (t1 == val) vote(tx1, tz1, gx, gz);
        }
        while (gm1) {
            const int b = __builtin_ctzll(gm1);
            gm1 &= gm1 - 1;
            const int val = __builtin_amdgcn_readlane(g1, b);
            const int c = b + 64;
            const int gx = c / 11, gz = c % 11;
            if (t0 == val) vote(tx0, tz0, gx, gz);
            if (t1 == val) vote(tx1, tz1, gx, gz);
        }
    }
    wave_sync();
    int best = 0;
    for (int w = lane; w < HIST_WORDS; w += WAVE) {
        const uint32_t v = hist[w];
        best = max(best, (int)max(v & 0xffff, v >> 16));
    }
    res.max_int = wave_max_i32(best);
    if (ARGMAX) {
        // argmax_intersection (tasks/task.py:121-136): first strict maximum in (rot, dx, dz) order
        int first = 0x7fffffff;
        if (res.max_int > 0) {
            for (int w = lane; w < HIST_WORDS; w += WAVE) {
                const uint32_t v = hist[w];
                if ((int)(v >> 16) == res.max_int) first = min(first, 2 * w + 1);
                if ((int)(v & 0xffff) == res.max_int) first = min(first, 2 * w);
            }
        }
        first = wave_min_i32(first);
        if (first != 0x7fffffff) {
            const int r = first / 121, rem = first % 121;
            res.arg_rot = r;
            const int xmax = r == 0 ? bb[0][1] : r == 1 ? bb[1][1] : r == 2 ? bb[2][1] : bb[3][1];
            const int zmax = r == 0 ? bb[0][3] : r == 1 ? bb[1][3] : r == 2 ? bb[2][3] : bb[3][3];
            res.arg_dx = rem / 11 + (xmax - 10);
            res.arg_dz = rem % 11 + (zmax - 10);
        }
    }
    wave_sync();
    return res;
}

// wave-wide copy of one 1104-byte row HBM -> LDS (69 aligned dwordx4)
__device__ inline void row_to_lds_wave(int8_t* dst_s, const int8_t* src_g) {
    const int lane = __lane_id();
    const uint4* s = reinterpret_cast<const uint4*>(src_g);
    uint4* d = reinterpret_cast<uint4*>(dst_s);
    for (int c = lane; c < CHUNKS; c += WAVE) d[c] = s[c];
}
}  // namespace igw
