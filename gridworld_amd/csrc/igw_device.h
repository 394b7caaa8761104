// igw_device.h -- device-side building blocks of the gridworld step path (gfx950, wave64).
//
// Execution model: GS lanes of a wavefront (GS = 64, 32, ..., 1; a "group") own one env.
// The serial double-precision physics chain is evaluated redundantly by every lane of the
// group (no broadcast needed, the wave issues the instruction anyway); the lanes of a group
// split the 40 ray-march samples of hit_test.  The per-step working set of an env is its
// 144-byte occupancy bitmap, staged in LDS; the int8 colour grid stays in HBM and is touched
// only where a colour matters.  Row-sized or rare work (histogram updates, rescans, resets,
// Task.__init__) is done by the whole wave, coalesced.
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
constexpr int BLOCK = 256;
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
// occupancy bitmap of the 1089 cells (bit = cell index): the per-step working set of the physics
constexpr int OCC_WORDS = IGW_OCC_WORDS;  // 36 dwords = 144 B per env in HBM
constexpr int OCC_PITCH = 37;             // LDS pitch (odd => conflict-free when one lane owns one env)
constexpr int HIST_ROW = IGW_HIST_ROW;    // persistent per-env vote histogram: 512 x u16 (484 used)

// gridworld/utils.py:9-24 and core/world.py:9
constexpr double WALKING_SPEED = 5.0;
constexpr double FLYING_SPEED = 15.0;
constexpr double GRAVITY = 20.0;
constexpr double JUMP_SPEED = 0x1.bb67ae8584caap+2;  // sqrt(2 * 20.0 * 1.2) = 6.928203230275509
constexpr double TERMINAL_VELOCITY = 50.0;
constexpr double PAD = 0.25;
constexpr double PI_OVER_180 = 0x1.1df46a2529d39p-6;   // pi / 180  (CPython math.radians)
constexpr double D180_OVER_PI = 0x1.ca5dc1a63c1f8p+5;  // 180 / pi  (CPython math.degrees)

struct alignas(16) AgentRec {  // layout documented in include/igw.h
    double x, y, z, yaw, pitch, vy;
    uint16_t step_no;
    int16_t size, prev_size, max_int;
    // bytes 0..5 inventory (int8); bits 48-49 time_int_steps code (2,4,8,12), bits 50-52 active_block,
    // bits 53-63 target_size of the synthetic task (copied from the task table by reset)
    uint64_t inv_pack;
};
static_assert(sizeof(AgentRec) == IGW_AGENT_BYTES, "agent record layout");

struct alignas(16) TaskMeta {
    double pose[5];
    int16_t target_size, env_max_int;
    uint8_t has_start;
    uint8_t pad0[3];
    int8_t bbox[16];  // offset 48: one aligned dwordx4
    int8_t inv_init[6];
    uint8_t pad[IGW_TASK_META_BYTES - 70];
};
static_assert(offsetof(TaskMeta, bbox) == 48, "bbox must be 16-byte aligned");
static_assert(sizeof(TaskMeta) == IGW_TASK_META_BYTES, "task meta layout");

struct KParams {
    int32_t n_envs, select_and_place, size_reward, max_steps, autoreset;
    int32_t debug;  // timing-only ablation switches (igw_config.reserved); 0 in every parity / bench run
    int32_t sample_tasks, n_tasks;  // igw_set_task_sampling: draw env_task uniformly from the table at every reset
    unsigned long long sample_seed;
    long long tick;                 // launches so far (keys the sampler)
    double right_scale, wrong_scale;
    int8_t* grid;
    uint32_t* occ;
    uint16_t* hist;
    AgentRec* agent;
    int32_t* env_task;
    const int8_t* task_target;
    const int8_t* task_start;
    const uint32_t* task_start_occ;
    const TaskMeta* task_meta;
    float* agent_pos;
    float* inventory;
    float* compass;
    float* reward;
    uint8_t* done;
    unsigned long long* stats;
    unsigned long long* stamps;  // diagnostic builds only: [waves][8] s_memtime stamps (igw_debug_set_stamps)
};

// ---------------------------------------------------------------- lane groups

// Fixed cross-lane patterns inside a quad (4 consecutive lanes) as DPP moves: one VALU instruction instead
// of a trip through the LDS crossbar (ds_bpermute).  CTRL = quad_perm: bits [2i+1:2i] = source lane of lane i.
template <int CTRL>
__device__ inline int dpp_quad(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
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
// uniform task index in [0, n): CustomTasks.reset() (gridworld/tasks/task_set.py:53-56) on the device
__host__ __device__ inline int rng_task(uint64_t seed, uint64_t env, uint64_t tick, int n) {
    uint64_t h = splitmix64(seed ^ splitmix64(env * 0x9E3779B1ull + tick * 0x100000001B3ull + 0x7461736bull));
    return (int)(((h >> 32) * (uint64_t)n) >> 32);
}
__host__ __device__ inline int rng_action18(uint64_t seed, uint64_t env, uint64_t t) {
    uint64_t h = splitmix64(seed ^ splitmix64(env * 0x100000001B3ull + t));
    return (int)(((h >> 32) * 18ull) >> 32);
}

// ---------------------------------------------------------------- env registers

struct Env {  // uniform across the lanes of a group
    double x, y, z, yaw, pitch, vy;
    int step_no, size, prev_size, max_int;
    uint64_t inv;  // 6 x int8
    int tis, active, target_size;
    int dirty;  // the histogram changed since max_int was last refreshed (a change with wrong_placement == 0)
};

__device__ inline int inv_get(uint64_t inv, int i) { return (int)(int8_t)(inv >> (8 * i)); }
__device__ inline uint64_t inv_add(uint64_t inv, int i, int d) {
    int v = inv_get(inv, i) + d;
    int sh = 8 * i;
    return (inv & ~(0xffull << sh)) | ((uint64_t)(uint8_t)v << sh);
}

__device__ inline void env_load(Env& e, const AgentRec* rec) {
    // every lane reads the same 64 B line (one request per wave)
    const AgentRec r = *rec;
    e.x = r.x; e.y = r.y; e.z = r.z; e.yaw = r.yaw; e.pitch = r.pitch; e.vy = r.vy;
    e.step_no = r.step_no; e.size = r.size; e.prev_size = r.prev_size & 0x7fff; e.max_int = r.max_int;
    e.dirty = ((uint16_t)r.prev_size >> 15) & 1;
    e.inv = r.inv_pack & 0x0000ffffffffffffull;
    const int code = (int)((r.inv_pack >> 48) & 3);
    e.tis = code == 0 ? 2 : code == 1 ? 4 : code == 2 ? 8 : 12;
    e.active = (int)((r.inv_pack >> 50) & 7);
    e.target_size = (int)(r.inv_pack >> 53);
}
__device__ inline void env_store(const Env& e, AgentRec* rec) {
    AgentRec r;
    r.x = e.x; r.y = e.y; r.z = e.z; r.yaw = e.yaw; r.pitch = e.pitch; r.vy = e.vy;
    r.step_no = (uint16_t)e.step_no; r.size = (int16_t)e.size;
    r.prev_size = (int16_t)(uint16_t)((e.prev_size & 0x7fff) | (e.dirty << 15)); r.max_int = (int16_t)e.max_int;
    const uint64_t code = e.tis == 2 ? 0 : e.tis == 4 ? 1 : e.tis == 8 ? 2 : 3;
    r.inv_pack = e.inv | (code << 48) | ((uint64_t)(e.active & 7) << 50) | ((uint64_t)(e.target_size & 0x7ff) << 53);
    *rec = r;
}

// ---------------------------------------------------------------- world queries

// core/world.py:57-58
__device__ inline bool build_zone_d(double x, double y, double z, double pad) {
    // -5-pad <= x <= 5+pad  <=>  |x| <= 5+pad (exact); likewise z
    return __builtin_fabs(x) <= 5.0 + pad && __builtin_fabs(z) <= 5.0 + pad && -1.0 - pad <= y && y < 8.0 + pad;
}
__device__ inline bool build_zone_i(int x, int y, int z) {
    return (unsigned)(x + 5) <= 10u && (unsigned)(z + 5) <= 10u && (unsigned)(y + 1) <= 8u;
}
__device__ inline int cell_of(int x, int y, int z) { return (y + 1) * LEVEL + (x + 5) * 11 + (z + 5); }

// `key in world` answered from the env's occupancy bitmap in LDS plus the fixed ground plane
// (World._initialize, core/world.py:60-71: y=-2, |x|,|z|<=18, WHITE(-1) over the build zone else GREY(0)).
// Colours live in the int8 grid row in HBM and are only fetched for the one block a break hits.
__device__ inline bool occ_test(const uint32_t* occ_s, int idx) { return (occ_s[idx >> 5] >> (idx & 31)) & 1u; }

__device__ inline bool world_has(const uint32_t* occ_s, int x, int y, int z) {
    if (y == -2) return x >= -18 && x <= 18 && z >= -18 && z <= 18;
    if (!build_zone_i(x, y, z)) return false;
    return occ_test(occ_s, cell_of(x, y, z));
}
// same without control flow (the bitmap word is always read, index clamped to 0 outside the zone): used
// where the lanes of a wave disagree on the outcome anyway and branches only cost exec-mask bookkeeping
__device__ inline bool world_has_nobranch(const uint32_t* occ_s, int x, int y, int z) {
    const bool ground = y == -2 && (unsigned)(x + 18) <= 36u && (unsigned)(z + 18) <= 36u;
    const bool in = build_zone_i(x, y, z);
    return (int)ground | ((int)in & (int)occ_test(occ_s, in ? cell_of(x, y, z) : 0));
}

// ---------------------------------------------------------------- trig front-end

struct TrigCtx {
    const double* lut;  // LDS copy of IGW_TRIG_LUT: {cos, sin} of radians(5k), k = IGW_LUT_K0..
};

// cos / sin of math.radians(deg).  Multiples of 5 degrees inside the table (all that discrete
// walking can produce) come from the LUT; anything else from the build's own
// double-precision sincos (igw_trig.h).
__device__ inline void sincos_deg(const TrigCtx& t, double deg, double& s, double& c) {
    int id = (int)deg;
    if ((double)id == deg && id >= 5 * IGW_LUT_K0 && id <= 5 * (IGW_LUT_K0 + IGW_LUT_N - 1) && (id % 5) == 0) {
        int k = id / 5 - IGW_LUT_K0;
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
// quotient (Markstein's theorem; checked bit-for-bit against x / 5.0 on 4*10^8 arguments).  Zeros and
// values near the underflow range (where the residual may be inexact) take the real division.
__device__ inline double div5(double x) {
    if (__builtin_fabs(x) >= 0x1p-900) {
        const double c = 0x1.999999999999ap-3;
        const double q = x * c;
        const double r = __builtin_fma(-5.0, q, x);
        return __builtin_fma(r, c, q);
    }
    return x / 5.0;
}

// ---------------------------------------------------------------- collide (core/world.py:264-310)

// Neighbourhood prober for collide: the 12 probes are np + small offsets, so the range checks of
// world_has (build zone per axis, ground plane y == -2 inside |x|,|z| <= 18) are hoisted per axis value
// and a probe is one add + one LDS bit test.
struct Probe {
    const uint32_t* occ_s;
    int base;            // cell_of(nx, ny, nz), may be out of range; only used when the axes are valid
    int nx, ny, nz;
    __device__ bool at(int dx, int dy, int dz) const {
        const int x = nx + dx, y = ny + dy, z = nz + dz;
        if (y == -2) return x >= -18 && x <= 18 && z >= -18 && z <= 18;
        const bool in = (unsigned)(x + 5) <= 10u && (unsigned)(z + 5) <= 10u && (unsigned)(y + 1) <= 8u;
        return in && occ_test(occ_s, base + dy * LEVEL + dx * 11 + dz);
    }
    // same without control flow: the bitmap word is always read (index clamped to 0 when out of range)
    __device__ bool at_nobranch(int dx, int dy, int dz) const {
        const int x = nx + dx, y = ny + dy, z = nz + dz;
        const bool ground = y == -2 && (unsigned)(x + 18) <= 36u && (unsigned)(z + 18) <= 36u;
        const bool in = (unsigned)(x + 5) <= 10u && (unsigned)(z + 5) <= 10u && (unsigned)(y + 1) <= 8u;
        const int idx = in ? base + dy * LEVEL + dx * 11 + dz : 0;
        return (int)ground | ((int)in & (int)occ_test(occ_s, idx));
    }
};

// Six faces in the reference order; a face only probes when its overlap test passes.  (Issuing all 12
// probes up front -- they depend only on np -- was measured slower at every group size.)
__device__ inline void collide(Env& e, const uint32_t* occ_s, double& px, double& py, double& pz) {
    const int nx = rint_i32(px), ny = rint_i32(py), nz = rint_i32(pz);
    const Probe w{occ_s, cell_of(nx, ny, nz), nx, ny, nz};
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
// results (and dy, which only the y faces clear) are then exchanged inside the group.
template <int GS>
__device__ inline void collide_split(const Grp<GS>& G, Env& e, const uint32_t* occ_s, double& px, double& py,
                                     double& pz) {
    static_assert(GS >= 4, "needs three lanes per env");
    const int nx = rint_i32(px), ny = rint_i32(py), nz = rint_i32(pz);
    const Probe w{occ_s, cell_of(nx, ny, nz), nx, ny, nz};
    const int a = G.gl & 3;
    const bool ax = a == 1, az = a == 2;
    const int ux = ax ? 1 : 0, uz = az ? 1 : 0, uy = (ax || az) ? 0 : 1;
    double pa = ax ? px : az ? pz : py;
    const double na = (double)(ax ? nx : az ? nz : ny);
    double vy = e.vy;
    const int i1 = ax ? -1 : 1;          // first face of the axis: (0,1,0), (-1,0,0), (0,0,1)
    const double f1 = ax ? -1.0 : 1.0;
    // predicated instead of branched: lanes of a wave disagree on almost every one of these tests, so the
    // branches bought nothing and cost exec-mask bookkeeping; the selects keep the arithmetic identical
    const int i2 = -i1;                  // second face: (0,-1,0), (1,0,0), (0,0,-1)
    const double f2 = ax ? 1.0 : -1.0;
    const bool b1 = (int)w.at_nobranch(ux * i1, uy * i1, uz * i1) | (int)w.at_nobranch(ux * i1, uy * i1 - 1, uz * i1);
    const bool b2 = (int)w.at_nobranch(ux * i2, uy * i2, uz * i2) | (int)w.at_nobranch(ux * i2, uy * i2 - 1, uz * i2);
    double d = (pa - na) * f1;
    const bool h1 = !(d < PAD) && b1;
    pa = h1 ? pa - (d - PAD) * f1 : pa;
    d = (pa - na) * f2;
    const bool h2 = !(d < PAD) && b2;
    pa = h2 ? pa - (d - PAD) * f2 : pa;
    if (uy && (h1 || h2)) vy = 0.0;
    py = dpp_quad<QUAD_BCAST0>(pa);
    px = dpp_quad<QUAD_BCAST1>(pa);
    pz = dpp_quad<QUAD_BCAST2>(pa);
    e.vy = dpp_quad<QUAD_BCAST0>(vy);
}

// ---------------------------------------------------------------- hit_test (core/world.py:73-99)

struct Hit {
    bool hit, have_prev;
    int bx, by, bz;  // the block that was hit (ground when by == -2)
    int px, py, pz;  // `previous`: the last empty cell in front of it
};

// Sample k is position + k sequential additions of vector/5 (the reference's recurrence, so the
// rounding of every partial sum is reproduced); the 40 samples are split over the lanes of the group.
template <int GS>
__device__ inline Hit hit_test(const Grp<GS>& G, const uint32_t* occ_s, double x, double y, double z,
                               double vx, double vy, double vz) {
    constexpr int SAMPLES = 40;  // max_distance 8 * m 5
    const double sx = div5(vx), sy = div5(vy), sz = div5(vz);  // dx / m with m = 5
    Hit h;
    h.hit = false; h.have_prev = false;
    h.bx = h.by = h.bz = 0;
    h.px = h.py = h.pz = 0;
    if constexpr (GS == 1) {
        // one lane per env: the reference loop as is; stop when every active lane has its answer
        int qx = 0, qy = 0, qz = 0;
        for (int s = 0; s < SAMPLES; s++) {
            const int kx = rint_i32(x), ky = rint_i32(y), kz = rint_i32(z);
            const bool differs = (s == 0) || kx != qx || ky != qy || kz != qz;
            if (!h.hit && differs && world_has(occ_s, kx, ky, kz)) {
                h.hit = true;
                h.have_prev = s != 0;
                h.bx = kx; h.by = ky; h.bz = kz;
                h.px = qx; h.py = qy; h.pz = qz;
            }
            if (!__any(!h.hit)) break;
            qx = kx; qy = ky; qz = kz;
            x = x + sx; y = y + sy; z = z + sz;
        }
        return h;
    } else {
        if constexpr (GS <= 2) {
            // Lane j of the group owns the contiguous samples j*C .. j*C+C-1.  It first walks to its chunk
            // (j*C sequential adds -- the reference's recurrence, nothing can be skipped), then evaluates its C
            // samples; `previous` is lane-local except for the chunk's first sample (one shuffle), and the
            // first `key != previous and key in world` of the whole ray is a per-lane scan + one group minimum.
            constexpr int C = (SAMPLES + GS - 1) / GS;
            const int lead = G.gl * C;  // samples in front of this lane's chunk
            for (int i = 0; i < (GS - 1) * C && i < SAMPLES - 1; i++) {
                if (i < lead) { x = x + sx; y = y + sy; z = z + sz; }
            }
            int key[C];  // packed (kx+128) | (ky+128) << 8 | (kz+128) << 16
            bool inw[C];
    #pragma unroll
            for (int k = 0; k < C; k++) {
                const int kx = rint_i32(x), ky = rint_i32(y), kz = rint_i32(z);
                key[k] = (kx + 128) | ((ky + 128) << 8) | ((kz + 128) << 16);
                inw[k] = world_has_nobranch(occ_s, kx, ky, kz);
                if (k + 1 < C) { x = x + sx; y = y + sy; z = z + sz; }
            }
            int q = G.shfl_up1(key[C - 1]);  // last key of the previous lane's chunk = `previous` of our first sample
            int first = SAMPLES, fkey = 0, fprev = 0;
    #pragma unroll
            for (int k = C - 1; k >= 0; k--) {  // descending, so the earliest candidate wins
                const int s = lead + k;
                const int prev = (k == 0) ? q : key[k - 1];
                const bool cand = s < SAMPLES && ((s == 0) || key[k] != prev) && inw[k];
                if (cand) { first = s; fkey = key[k]; fprev = prev; }
            }
            int best = first;
    #pragma unroll
            for (int o = GS / 2; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, GS));
            if (best < SAMPLES) {
                const int owner = best / C;
                const int bk = G.bcast(fkey, owner), pk = G.bcast(fprev, owner);
                h.hit = true;
                h.have_prev = best != 0;
                h.bx = (bk & 0xff) - 128; h.by = ((bk >> 8) & 0xff) - 128; h.bz = ((bk >> 16) & 0xff) - 128;
                h.px = (pk & 0xff) - 128; h.py = ((pk >> 8) & 0xff) - 128; h.pz = ((pk >> 16) & 0xff) - 128;
            }
        } else {
            // Strided mapping (lane j owns samples j, j+GS, ...): every round after the first advances all
            // lanes by GS unmasked adds, which is cheaper than the chunked walk once groups are 4+ wide.
            constexpr int ROUNDS = (SAMPLES + GS - 1) / GS;
            // pass 1: every lane walks to its samples (pure ALU) and issues all its membership probes
            int key[ROUNDS];  // packed (kx+128) | (ky+128) << 8 | (kz+128) << 16
            bool inw[ROUNDS];
    #pragma unroll
            for (int r = 0; r < ROUNDS; r++) {
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
                const int kx = rint_i32(x), ky = rint_i32(y), kz = rint_i32(z);
                key[r] = (kx + 128) | ((ky + 128) << 8) | ((kz + 128) << 16);
                inw[r] = world_has_nobranch(occ_s, kx, ky, kz);
            }
            // pass 2: `key != previous and key in world`, first sample wins.  Every lane scans its own samples
            // (previous = the neighbouring lane's key of the same round, or the last lane's key of the round
            // before), then one group minimum picks the earliest candidate: no ballot / branch per round.
            int first = SAMPLES, fkey = 0, fprev = 0;
#pragma unroll
            for (int r = ROUNDS - 1; r >= 0; r--) {  // descending, so the earliest candidate wins
                const int s = r * GS + G.gl;
                int q = G.shfl_up1(key[r]);
                if (r > 0) {
                    const int wrap = G.bcast_last(key[r - 1]);
                    if (G.gl == 0) q = wrap;
                }
                const bool cand = s < SAMPLES && ((s == 0) || key[r] != q) && inw[r];
                if (cand) { first = s; fkey = key[r]; fprev = q; }
            }
            const int best = G.group_min(first);
            if (best < SAMPLES) {
                const int owner = best % GS;
                const int bk = G.bcast(fkey, owner), pk = G.bcast(fprev, owner);
                h.hit = true;
                h.have_prev = best != 0;
                h.bx = (bk & 0xff) - 128; h.by = ((bk >> 8) & 0xff) - 128; h.bz = ((bk >> 16) & 0xff) - 128;
                h.px = (pk & 0xff) - 128; h.py = ((pk >> 8) & 0xff) - 128; h.pz = ((pk >> 16) & 0xff) - 128;
            }
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
