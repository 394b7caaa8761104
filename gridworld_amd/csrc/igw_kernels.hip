// igw_kernels.hip -- HIP kernels (gfx950) and the C ABI of include/igw.h.
//
// Reference semantics reproduced (file:line relative to iglu-contest/gridworld):
//   World.step                gridworld/core/world.py:434-456
//   movement / move_camera    gridworld/core/world.py:338-356
//   place_or_remove_block     gridworld/core/world.py:312-332  (+ env.py:136-153 grid callbacks)
//   update / _update          gridworld/core/world.py:203-262
//   GridWorld.step tail       gridworld/env.py:276-303, SizeReward.step env.py:325-331
//   Task.step_intersection    gridworld/tasks/task.py:103-119
//   GridWorld.reset           gridworld/env.py:206-261
//   Task.__init__             gridworld/tasks/task.py:9-72
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see gridworld_amd/build.py).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "igw_device.h"
#include "igw_trig_lut.h"

namespace igw {

enum { MODE_WALK = 0, MODE_FLY = 1, MODE_WALK_DICT = 2 };

struct ActIn {
    const int32_t* actions;
    const float* movement;
    const float* camera;
    const int32_t* inventory;
    const int32_t* placement;
    const uint8_t* buttons;  // walking Dict: [N][8] forward, back, left, right, jump, attack, use, hotbar
};

struct StepOut {
    double reward;
    bool done;
};

// In-kernel phase stamps exist only in the diagnostic build of the library (-DIGW_DIAG, libigw_hip_diag.so,
// tools/stamp_phases.py): drain this wave's memory queues so the time is charged to the phase that caused
// the wait, then read the shader clock.  The production library compiles them (and the timing-only
// ablation switches of igw_config.reserved) out.
#ifdef IGW_DIAG
#define IGW_DIAG_FLAG(p, bit) ((p).debug & (bit))
__device__ inline void stamp(const KParams& p, int slot) {
    if (p.stamps) {
        // debug bit 8: do not drain -- the stamp then marks when the wave REACHES this point, with the loads of
        // earlier phases still in flight exactly as in the production kernel
        if (!(p.debug & 8)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (__lane_id() == 0) p.stamps[((size_t)blockIdx.x * WAVES_PER_BLOCK + threadIdx.x / WAVE) * 8 + slot] = t;
    }
}
// slot 7: what this wave had to do (changed envs, rescans, resets, longest sub-step count), for tail analysis
__device__ inline void stamp_features(const KParams& p, unsigned long long v) {
    if (p.stamps && __lane_id() == 0) p.stamps[((size_t)blockIdx.x * WAVES_PER_BLOCK + threadIdx.x / WAVE) * 8 + 7] = v;
}
#else
#define IGW_DIAG_FLAG(p, bit) 0
__device__ inline void stamp(const KParams&, int) {}
__device__ inline void stamp_features(const KParams&, unsigned long long) {}
#endif

// Counter atomics go out as GLOBAL atomics without return.  Through a generic pointer the compiler emits flat_atomic and,
// because a flat access may alias LDS, puts an s_waitcnt vmcnt(0) in front of it whenever global stores are in flight:
// at the end of a step that waited for the acknowledgement of every store of the wavefront (histogram rows, records) --
// 700 cycles on every wavefront with something to count.
// IGW_STEPS_MODE: how a step launch adds its env-steps to IGW_STAT_STEPS (2: once per launch; 1: once per block --
// round 5; 0: not at all -- round 4; the other two exist for the same-box A/B of tools/ab_variants.sh)
// IGW_SPLIT_BURST: the step kernel waits for its small input loads first and for the occupancy pieces behind the
// action parse / trig (see step_kernel); 0 = one wait for the whole burst (rounds 3-5).  Same-box A/B
// (profiles/r06_ab_variants.txt): -0.4 % walking, -1.1 % flying (its trig runs while the occupancy rows arrive).
#ifndef IGW_SPLIT_BURST
#define IGW_SPLIT_BURST 1
#endif
#ifndef IGW_EARLY_COUNTERS
#define IGW_EARLY_COUNTERS 1
#endif
#ifndef IGW_STEPS_MODE
#define IGW_STEPS_MODE 2
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) unsigned long long global_u64;
__device__ inline void counter_add(unsigned long long* p, unsigned long long v) {
    __hip_atomic_fetch_add((global_u64*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
__device__ inline void counter_add(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
#endif
__device__ inline void stat_add(unsigned long long* stats, int which, unsigned long long v) {
    if (stats) counter_add(&stats[(blockIdx.x & (IGW_STAT_STRIPES - 1)) * 8 + which], v);
}

struct BBox {
    int xmin, xmax, zmin, zmax;
};

// bounding box of the non-zero cells of an LDS row (wave-wide); empty -> (10, 0, 10, 0)
__device__ inline BBox row_bbox(const int8_t* row_s, int& nnz) {
    const int lane = __lane_id();
    int xmin = 10, xmax = 0, zmin = 10, zmax = 0, cnt = 0;
    for (int c = lane; c < CELLS; c += WAVE) {
        if (row_s[c] != 0) {
            const int r = c % LEVEL, x = r / 11, z = r % 11;
            xmin = min(xmin, x); xmax = max(xmax, x);
            zmin = min(zmin, z); zmax = max(zmax, z);
            cnt++;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    nnz = cnt;
    BBox b;
    b.xmin = wave_min_i32(xmin); b.xmax = wave_max_i32(xmax);
    b.zmin = wave_min_i32(zmin); b.zmax = wave_max_i32(zmax);
    if (cnt == 0) { b.xmin = 10; b.xmax = 0; b.zmin = 10; b.zmax = 0; }
    return b;
}

__device__ inline int pack_bbox(int xmin, int xmax, int zmin, int zmax) {
    return (xmin & 0xff) | ((xmax & 0xff) << 8) | ((zmin & 0xff) << 16) | ((zmax & 0xff) << 24);
}
// bboxes of the 4 rotations (x,z) -> (z, 10-x) (tasks/task.py:52-53) of a box; admissible translations
// of rotation r are dx in [xmax-10, xmin], dz in [zmax-10, zmin] (tasks/task.py:62-72 as a bbox rule)
__device__ inline void rot_bboxes(const BBox& b, bool empty, int* out4) {
    if (empty) {
        out4[0] = out4[1] = out4[2] = out4[3] = pack_bbox(10, 0, 10, 0);
        return;
    }
    out4[0] = pack_bbox(b.xmin, b.xmax, b.zmin, b.zmax);
    out4[1] = pack_bbox(b.zmin, b.zmax, 10 - b.xmax, 10 - b.xmin);
    out4[2] = pack_bbox(10 - b.xmax, 10 - b.xmin, 10 - b.zmax, 10 - b.zmin);
    out4[3] = pack_bbox(10 - b.zmax, 10 - b.zmin, b.xmin, b.xmax);
}

// What GridWorld.reset takes from the task's metadata row, as registers.  A wavefront that knows at the START of a
// step that one of its episodes runs out in it (step_no + 1 == max_steps, the usual end of an episode) fetches
// these with the step's other inputs instead of at the end, where two dependent memory round trips -- each behind an
// s_waitcnt that also waits for the step's stores -- used to make the resetting wavefront the last of its CU.
struct ResetVals {
    double pose[5];
    int target_size;
    uint32_t inv01, inv23, inv45;  // env.py:243-246: 20 - blocks of the colour in the starting grid
    bool has_start;
};
struct ResetMeta {   // bytes 0..47 and 64..79 of the metadata row, RAW: nothing is converted where it is loaded,
    vu4 a, b, c, d;    // so nothing waits for these loads before reset_decode() -- the row's layout is in include/igw.h
};
__device__ inline ResetMeta load_reset_meta(const TaskMeta* meta) {
    const vu4* m = reinterpret_cast<const vu4*>(meta);
    return ResetMeta{gload(m), gload(m + 1), gload(m + 2), gload(m + 4)};
}
// "Any value" registers for the lanes that do not prefetch (any_value(): a register without an instruction): the merge
// of a lane-masked load with such a value needs no copy, so the compiler neither waits for the loads where the branch
// ends nor fills sixteen registers with zeros in every wavefront (what `ResetMeta pre = {}` cost).
__device__ inline ResetMeta reset_meta_any() {
    ResetMeta r;
    r.a = vu4{any_value_u(), any_value_u(), any_value_u(), any_value_u()};
    r.b = vu4{any_value_u(), any_value_u(), any_value_u(), any_value_u()};
    r.c = vu4{any_value_u(), any_value_u(), any_value_u(), any_value_u()};
    r.d = vu4{any_value_u(), any_value_u(), any_value_u(), any_value_u()};
    return r;
}
__device__ inline double u2d(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ inline ResetVals reset_decode(const ResetMeta& r) {
    ResetVals v;
    v.pose[0] = u2d(r.a.x, r.a.y); v.pose[1] = u2d(r.a.z, r.a.w);
    v.pose[2] = u2d(r.b.x, r.b.y); v.pose[3] = u2d(r.b.z, r.b.w);
    v.pose[4] = u2d(r.c.x, r.c.y);
    v.target_size = (int)(int16_t)(r.c.z & 0xffffu);
    v.has_start = (r.c.w & 0xffu) != 0;
    v.inv01 = r.d.x; v.inv23 = r.d.y; v.inv45 = r.d.z;
    return v;
}
// per-lane part of GridWorld.reset (env.py:206-261): everything except the grid / bitmap rows.
// generated_size >= 0: the task row was just written by the on-device RandomTasks generator of this wave (its
// metadata is taken from registers, not re-read: target size as given, empty start => full inventory).
__device__ inline void reset_env_regs(Env& e, const ResetVals& m, bool keep_size, int generated_size = -1) {
    if (!keep_size) e.size = 0;  // SizeReward.reset, env.py:321-323
    e.step_no = 0;               // env.py:217
    e.prev_size = 0;             // _synthetic_task.reset(): prev_grid_size = 0, max_int = 0 (task.py:74-86)
    e.max_int = 0;
    e.dirty = 0;
    e.x = m.pose[0]; e.y = m.pose[1]; e.z = m.pose[2];  // env.py:239-240
    e.yaw = m.pose[3]; e.pitch = m.pose[4];
    if (generated_size >= 0) {   // the row was just written by the on-device generator: empty start, full inventory
        e.target_size = generated_size;
        e.inv01 = e.inv23 = e.inv45 = INV_FULL_PAIR;
    } else {
        e.target_size = m.target_size;
        e.inv01 = m.inv01; e.inv23 = m.inv23; e.inv45 = m.inv45;
    }
    // agent.dy, time_int_steps, active_block are NOT reset by the reference
}
__device__ inline void reset_env_regs(Env& e, const TaskMeta* meta, bool keep_size, int generated_size = -1) {
    reset_env_regs(e, reset_decode(load_reset_meta(meta)), keep_size, generated_size);
}

// A reset counts a new episode and, with a task generator on the device, picks the env's next task row.
// Returns the number of the episode that ends (the samplers' key).  Task and counter live in the env's aux record,
// which the caller stores.
__device__ inline uint32_t next_task(const KParams& p, int env, Env& e, int& task) {
    const uint32_t ep = e.episode;
    e.episode = ep + 1;
    if (p.sample_tasks) {  // CustomTasks.reset on the device: uniform choice over the table
        task = rng_task(p.sample_seed, (uint64_t)(p.env_base + env), (uint64_t)ep, p.n_tasks);
    } else if (p.rt_enabled) {  // RandomTasks.sample_task on the device: the env's own row is regenerated
        task = env;
    }
    return ep;
}

// The colour index of a synthetic target (see "incremental maximal_intersection" below; layout in include/igw.h):
// per y level a 160-byte block = the four rotation bounding boxes (16 B), offs[15] (start of each colour class
// in the cell list, offs[14] = number of target cells on the level), cells[<= 121] = (x << 4 | z) of the level's
// target cells sorted by colour class -- a counting sort by the whole wave (lane = cell, two cells per lane).
// tgt_s: LDS copy of the synthetic target row; stage_s: 160 bytes of LDS; out_g: the task's IGW_TASK_INDEX_BYTES.
__device__ inline int index_class(int c) { return (c == 0 || c < -7 || c > 7) ? -1 : (c < 0 ? c + 7 : c + 6); }
__device__ inline void build_level_index_wave(const int8_t* tgt_s, const int* bb4, uint8_t* stage_s, uint8_t* out_g) {
    const int lane = __lane_id();
    const int c0 = lane, c1 = lane + 64;
    const bool v1 = c1 < LEVEL;
    const uint32_t p0 = (uint32_t)((c0 / 11) << 4 | (c0 % 11)), p1 = (uint32_t)((c1 / 11) << 4 | (c1 % 11));
    uint4* st4 = reinterpret_cast<uint4*>(stage_s);
    for (int y = 0; y < IGW_GRID_Y; y++) {
        const int k0 = index_class(tgt_s[y * LEVEL + c0]), k1 = v1 ? index_class(tgt_s[y * LEVEL + c1]) : -1;
        if (lane < IGW_LEVEL_INDEX_BYTES / 16) st4[lane] = lane == 0 ? make_uint4((uint32_t)bb4[0], (uint32_t)bb4[1], (uint32_t)bb4[2], (uint32_t)bb4[3]) : make_uint4(0, 0, 0, 0);
        wave_sync();
        if (__ballot(k0 >= 0 || k1 >= 0)) {  // (most levels of most targets are empty: all offsets stay 0)
            int off = 0;
            for (int k = 0; k < 14; k++) {
                const uint64_t m0 = __ballot(k0 == k), m1 = __ballot(k1 == k);
                const int n0 = __builtin_popcountll(m0);
                if (lane == 0) stage_s[16 + k] = (uint8_t)off;
                if (k0 == k) stage_s[32 + off + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] = (uint8_t)p0;
                if (k1 == k) stage_s[32 + off + n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] = (uint8_t)p1;
                off += n0 + __builtin_popcountll(m1);
            }
            if (lane == 0) stage_s[16 + 14] = stage_s[16 + 15] = (uint8_t)off;   // (offs[15] = offs[14]: the empty slice change_class() clamps to)
        }
        wave_sync();
        if (lane < IGW_LEVEL_INDEX_BYTES / 16) reinterpret_cast<uint4*>(out_g + y * IGW_LEVEL_INDEX_BYTES)[lane] = st4[lane];
        wave_sync();
    }
}

// RandomTasks.sample_task (gridworld/tasks/task_set.py:135-157) for one env by the whole wave: per height level
// the first block uniform over the plane, then K = min(max_blocks - 1, free window cells) further blocks on
// distinct cells within Chebyshev distance max_dist of it.  The reference draws them one at a time by rejection,
// i.e. uniformly without replacement, so the occupied set is a uniformly random K-subset of the window; here
// every window cell gets a random 32-bit key and the K smallest win (ties by cell index).  Colours are iid
// uniform in 1..num_colors.  row_s: STRIDE + IGW_LEVEL_INDEX_BYTES bytes of LDS scratch.  Writes the task's target row, its colour index and the generated
// part of its metadata (the init pose is kept); returns the number of blocks (= target_size).
__device__ inline int sample_random_task_wave(const KParams& p, int env, uint32_t ep, int8_t* row_s) {
    const int lane = __lane_id();
    uint4* r4 = reinterpret_cast<uint4*>(row_s);
    for (int c = lane; c < CHUNKS; c += WAVE) r4[c] = make_uint4(0, 0, 0, 0);
    wave_sync();
    const unsigned long long genv = (unsigned long long)(p.env_base + env);
    uint32_t h0 = hash_combine((uint32_t)p.sample_seed, (uint32_t)(p.sample_seed >> 32));
    h0 = hash_combine(hash_combine(hash_combine(h0, (uint32_t)genv), (uint32_t)(genv >> 32)), ep);
    const int d = p.rt_max_dist, ncol = p.rt_colors;
    int total = 0, xmin = 10, xmax = 0, zmin = 10, zmax = 0;
    for (int lvl = 0; lvl < p.rt_levels; lvl++) {
        const uint32_t hl = hash_combine(h0, 0x1000u + (uint32_t)lvl);
        const int bx = rng_below(hash_combine(hl, 0x20001u), IGW_GRID_X), bz = rng_below(hash_combine(hl, 0x20002u), IGW_GRID_Z);
        const int x0 = max(bx - d, 0), x1 = min(bx + d, IGW_GRID_X - 1), z0 = max(bz - d, 0), z1 = min(bz + d, IGW_GRID_Z - 1);
        const int K = min(p.rt_max_blocks - 1, (x1 - x0 + 1) * (z1 - z0 + 1) - 1);
        // this lane's two plane cells
        const int c0 = lane, c1 = lane + 64;
        const int cx0 = c0 / 11, cz0 = c0 % 11, cx1 = c1 / 11, cz1 = c1 % 11;
        const bool w0 = cx0 >= x0 && cx0 <= x1 && cz0 >= z0 && cz0 <= z1 && !(cx0 == bx && cz0 == bz);
        const bool w1 = c1 < LEVEL && cx1 >= x0 && cx1 <= x1 && cz1 >= z0 && cz1 <= z1 && !(cx1 == bx && cz1 == bz);
        const uint32_t k0 = hash_combine(hl, (uint32_t)c0), k1 = hash_combine(hl, (uint32_t)c1);
        int rank0 = 0, rank1 = 0;
        for (int x = x0; x <= x1; x++) {
            for (int z = z0; z <= z1; z++) {
                const int j = x * 11 + z;
                if (x == bx && z == bz) continue;
                const uint32_t kj = hash_combine(hl, (uint32_t)j);
                rank0 += (kj < k0 || (kj == k0 && j < c0)) ? 1 : 0;
                rank1 += (kj < k1 || (kj == k1 && j < c1)) ? 1 : 0;
            }
        }
        const bool s0 = w0 && rank0 < K, s1 = w1 && rank1 < K;
        const bool f0 = cx0 == bx && cz0 == bz, f1 = c1 < LEVEL && cx1 == bx && cz1 == bz;  // the first block
        if (s0 || f0) row_s[lvl * LEVEL + c0] = (int8_t)(1 + rng_below(hash_combine(hl, 0x10000u + (uint32_t)c0), ncol));
        if (s1 || f1) row_s[lvl * LEVEL + c1] = (int8_t)(1 + rng_below(hash_combine(hl, 0x10000u + (uint32_t)c1), ncol));
        total += K + 1;
        int lxmin = 10, lxmax = 0, lzmin = 10, lzmax = 0;
        if (s0 || f0) { lxmin = cx0; lxmax = cx0; lzmin = cz0; lzmax = cz0; }
        if (s1 || f1) { lxmin = min(lxmin, cx1); lxmax = max(lxmax, cx1); lzmin = min(lzmin, cz1); lzmax = max(lzmax, cz1); }
        xmin = min(xmin, wave_min_i32(lxmin)); xmax = max(xmax, wave_max_i32(lxmax));
        zmin = min(zmin, wave_min_i32(lzmin)); zmax = max(zmax, wave_max_i32(lzmax));
    }
    wave_sync();
    uint4* dst = reinterpret_cast<uint4*>(const_cast<int8_t*>(p.task_target) + (size_t)env * STRIDE);
    for (int c = lane; c < CHUNKS; c += WAVE) dst[c] = r4[c];
    {   // the colour index of the new target (what the step kernels vote from)
        BBox bi;
        bi.xmin = xmin; bi.xmax = xmax; bi.zmin = zmin; bi.zmax = zmax;
        int bbi[4];
        rot_bboxes(bi, total == 0, bbi);
        build_level_index_wave(row_s, bbi, reinterpret_cast<uint8_t*>(row_s) + STRIDE,
                               const_cast<uint8_t*>(p.task_index) + (size_t)env * IGW_TASK_INDEX_BYTES);
    }
    if (lane == 0) {
        // bytes 40..69 of the metadata row (task.py:9-72 on an empty start): target_size, GridWorld.max_int = 0,
        // has_start = 0, the four rotation bounding boxes, inventory 20 x 6
        TaskMeta* m = const_cast<TaskMeta*>(p.task_meta) + env;
        BBox b;
        b.xmin = xmin; b.xmax = xmax; b.zmin = zmin; b.zmax = zmax;
        int bb[4];
        rot_bboxes(b, total == 0, bb);
        m->target_size = (int16_t)total;
        m->env_max_int = 0;
        m->has_start = 0;
        for (int k = 0; k < 4; k++) {
            m->bbox[4 * k + 0] = (int8_t)(bb[k] & 0xff);
            m->bbox[4 * k + 1] = (int8_t)((bb[k] >> 8) & 0xff);
            m->bbox[4 * k + 2] = (int8_t)((bb[k] >> 16) & 0xff);
            m->bbox[4 * k + 3] = (int8_t)((bb[k] >> 24) & 0xff);
        }
        for (int k = 0; k < 6; k++) m->inv_init[k] = 20;
    }
    wave_sync();
    return total;
}


// whole-wave copy of the starting grid and its occupancy bitmap into one env's rows
// (env.py:234-238: world := starting grid); occ_s (the env's LDS occupancy row) may be nullptr when the
// kernel ends right after
__device__ inline void reset_rows_wave(const KParams& p, int env, int task, bool has_start, uint32_t* occ_s) {
    const int lane = __lane_id();
    uint4* dg = reinterpret_cast<uint4*>(p.grid + (size_t)env * STRIDE);
    uint4* dh = reinterpret_cast<uint4*>(p.hist + (size_t)env * HIST_ROW);
    const uint4 zero = make_uint4(0, 0, 0, 0);
    if (!has_start) {   // (wave-uniform) no starting grid: stores only -- nothing to load, so nothing to wait for
        for (int c = lane; c < CHUNKS; c += WAVE) gstore(dg + c, zero);
        if (lane < OCC_WORDS) {
            gstore(p.occ + (size_t)env * OCC_WORDS + lane, 0u);
            if (occ_s) occ_s[OCC_VAR0 + lane] = 0u;
        }
        gstore(dh + lane, zero);
        return;
    }
    const uint4* sg = reinterpret_cast<const uint4*>(p.task_start + (size_t)task * STRIDE);
    // (global address space throughout: these pointers came through a scalar-register barrier and are generic to the
    // compiler, which waits for every memory operation in flight in front of a flat one)
    for (int c = lane; c < CHUNKS; c += WAVE) gstore(dg + c, has_start ? gload(sg + c) : zero);
    if (lane < OCC_WORDS) {
        const uint32_t v = has_start ? gload(p.task_start_occ + (size_t)task * OCC_WORDS + lane) : 0u;
        gstore(p.occ + (size_t)env * OCC_WORDS + lane, v);
        if (occ_s) occ_s[OCC_VAR0 + lane] = v;
    }
    // grid == start  =>  synthetic grid empty  =>  no votes
    gstore(dh + lane, zero);
}

// The same for the ONE env per wavefront whose starting row the step fetched ahead (step_kernel, "staged"): the grid row is
// in LDS, the occupancy words in a register of lanes 0..47 -- stores only, no memory round trip at the end of the step.
__device__ inline void reset_rows_staged(const KParams& p, int env, const uint32_t* row_s, const uint32_t* tail_s, uint32_t occ_w) {
    const int lane = __lane_id();
    uint4* dg = reinterpret_cast<uint4*>(p.grid + (size_t)env * STRIDE);
    gstore(dg + lane, reinterpret_cast<const uint4*>(row_s)[lane]);
    if (lane < CHUNKS - WAVE) gstore(dg + WAVE + lane, reinterpret_cast<const uint4*>(tail_s)[lane]);
    if (lane < OCC_WORDS) gstore(p.occ + (size_t)env * OCC_WORDS + lane, occ_w);
    gstore(reinterpret_cast<uint4*>(p.hist + (size_t)env * HIST_ROW) + lane, make_uint4(0, 0, 0, 0));
}

// The agent record as its four 16-byte pieces, lane q of the env's (first) quad stores piece q: ONE store instruction
// for the whole record.  (By value: a select between FIELDS of the env struct becomes a load from a selected address
// and parks the struct in scratch memory.)
__device__ __forceinline__ uint4 agent_piece(int q, double x, double y, double z, double yaw, double pitch, double vy, const uint4& piece3) {
    const double a = q == 0 ? x : q == 1 ? z : pitch, b = q == 0 ? y : q == 1 ? yaw : vy;
    const uint4 pose = make_uint4((uint32_t)__double2loint(a), (uint32_t)__double2hiint(a), (uint32_t)__double2loint(b), (uint32_t)__double2hiint(b));
    return q == 3 ? piece3 : pose;
}

struct CellChange {
    int idx;  // dense cell index; -1: grid unchanged this step
    int lvl;  // ... its y level (idx / 121) and (x + 5) | (z + 5) << 4 within the level: known where the cell is chosen,
    int xz;   //     so nothing downstream divides
    int bit;  // the cell's bit in the HBM occupancy row
    // old_val: for a break, the cell's colour before the step as the RAW zero-extended byte of a (possibly still pending)
    // load: converting it where it is loaded would make the wave wait for the load there; old_colour() does it at the
    // use.  Any value for the lanes that do not break.
    int old_val, new_val;
    uint32_t occ_word;  // the changed word of the occupancy row, as updated (leader lane; goes back to HBM at the end)
};
// (a placement fills a cell that was empty; only a break has loaded the colour.  The empty asm statement keeps the
// conversion of the raw byte HERE: left alone the compiler hoists it to the load, and with it the wait for the load.)
__device__ inline int old_colour(const CellChange& ch) {
    int raw = ch.old_val;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(raw));
#endif
    return ch.new_val != 0 ? 0 : (int)(int8_t)raw;
}

struct Motion {  // get_motion_vector (core/world.py:163-201): constant over the sub-steps of one step
    double x, y, z;
};

// World.step (core/world.py:434-456) after action parsing, first half: movement, camera, place / break -- in two
// parts with the ray march (hit_test) between them: world_act_pre needs nothing of the occupancy row, hit_test and
// world_act_post do.
struct ActPre {
    bool want_sight, add, remove;
    double vx, vy, vz;  // sight vector (get_sight_vector, core/world.py:312-318); only when want_sight
};

// movement, move_camera, the step's trig and the motion vector.  Every lane of the group runs it with identical
// inputs.
template <int GS, int MODE>
__device__ inline ActPre world_act_pre(const Grp<GS>& G, const KParams& p, Env& e, const TrigCtx& trig, double s0, double s1,
                                       double dy, int inventory, double cam0, double cam1, bool remove, bool add, Motion& mv) {
    constexpr bool FLY = MODE == MODE_FLY;
    if (p.select_and_place && inventory != 0) { add = true; remove = false; }  // :444-446
    // movement, :344-356
    e.vy = ((dy != 0.0) & (e.vy == 0.0)) ? JUMP_SPEED * dy : e.vy;
    if (FLY) e.vy = dy == 0.0 ? 0.0 : e.vy;
    e.active = (unsigned)(inventory - 1) < 6u ? inventory : e.active;
    // move_camera, :338-342
    e.yaw = e.yaw + cam0;
    {
        double y = e.pitch + cam1;
        y = 90.0 < y ? 90.0 : y;
        y = -90.0 > y ? -90.0 : y;
        e.pitch = y;
    }
    // The step needs up to three sin/cos pairs, all of the post-camera rotation: pitch (sight vector and,
    // when flying, motion vector -- one evaluation serves both, same argument => same value), yaw - 90
    // (sight vector) and yaw + strafe heading (motion vector).  They are independent, so in groups of 4+
    // lanes each is evaluated by one lane (same code, argument chosen by lane) and exchanged.
    ActPre a;
    a.add = add; a.remove = remove;
    a.want_sight = add != remove && !IGW_DIAG_FLAG(p, 2);
    const bool want_sight = a.want_sight;
    const bool strafing = s0 != 0.0 || s1 != 0.0;
    double strafe_deg = 0.0;
    if constexpr (MODE == MODE_WALK) {
        // math.degrees(math.atan2(*agent.strafe)), :176.  Discrete(18) moves along one axis at a time: (s0, s1) =
        // (-1, 0) -> -90, (1, 0) -> 90, (0, -1) -> 180, (0, 1) -> 0, exactly -- as a select of the high word (the low
        // word of all four doubles is zero; nested ifs on doubles compile into exec-mask branches)
        // (integer arithmetic on the two signs: a chain of selects on double comparisons compiles into branches again)
        const int i0 = (int)s0, i1 = (int)s1;                       // -1, 0, 1
        const uint32_t hi = (((uint32_t)(i0 < 0) << 31)             // -90: the sign
                             | ((0u - (uint32_t)((i0 != 0) | (i1 < 0))) & 0x40568000u))   // 90 (and -90, 180): exponent and mantissa of 90
                            + ((uint32_t)(i1 < 0) << 20);           // 180 = 2 x 90: one more in the exponent field
        strafe_deg = __hiloint2double((int)hi, 0);
    } else if (strafing) {
        if (!FLY && s1 == 0.0) strafe_deg = s0 < 0.0 ? -90.0 : 90.0;       // degrees(atan2(-+1, 0))
        else if (!FLY && s0 == 0.0) strafe_deg = s1 < 0.0 ? 180.0 : 0.0;   // degrees(atan2(0, -+1))
        else strafe_deg = igw_atan2(s0, s1) * D180_OVER_PI;
    }
    const bool want_pitch = want_sight || (FLY && strafing);
    double sp = 0.0, cp = 1.0, sy = 0.0, cy = 1.0, sx = 0.0, cx = 1.0;
    if constexpr (GS >= 4) {
        const int l = G.gl & 3;
        const double deg = l == 1 ? e.yaw - 90.0 : l == 2 ? e.yaw + strafe_deg : e.pitch;
        const bool need = l == 1 ? want_sight : l == 2 ? strafing : want_pitch;
        double sv = 0.0, cv = 1.0;
        if (need) sincos_deg<!FLY>(trig, deg, sv, cv);
        sp = dpp_quad<QUAD_BCAST0>(sv); cp = dpp_quad<QUAD_BCAST0>(cv);
        sy = dpp_quad<QUAD_BCAST1>(sv); cy = dpp_quad<QUAD_BCAST1>(cv);
        sx = dpp_quad<QUAD_BCAST2>(sv); cx = dpp_quad<QUAD_BCAST2>(cv);
    } else {
        if (want_pitch) sincos_deg<!FLY>(trig, e.pitch, sp, cp);
        if (want_sight) sincos_deg<!FLY>(trig, e.yaw - 90.0, sy, cy);
        if (strafing) sincos_deg<!FLY>(trig, e.yaw + strafe_deg, sx, cx);
    }
    // get_motion_vector, :163-201 (rotation and strafe are constant over the sub-steps)
    mv.x = 0.0; mv.y = 0.0; mv.z = 0.0;
    if (strafing) {
        if (FLY) {
            double mm = cp;
            mv.y = sp;
            if (s1 != 0.0) { mv.y = 0.0; mm = 1.0; }
            if (s0 > 0.0) mv.y *= -1.0;
            mv.x = cx * mm;
            mv.z = sx * mm;
        } else {
            mv.x = cx;
            mv.z = sx;
        }
    }
    // m = cos(radians(y)); dy = sin(radians(y)); dx = cos(radians(x - 90)) * m; dz = sin(radians(x - 90)) * m
    a.vx = cy * cp; a.vy = sp; a.vz = sy * cp;
    return a;
}

// place_or_remove_block, :312-332, given the result of the ray march.  For a break, ch.old_val is the pending
// colour load of the block that was hit: nothing here waits for it (finish_break consumes it after the physics).
// FUSED: the caller is the fused loop, where an earlier step of the SAME launch may have written the row (the colour is
// then read past this CU's L1).
template <int GS, bool FUSED = true>
__device__ inline CellChange world_act_post(const Grp<GS>& G, Env& e, uint32_t* occ_s, const int8_t* grid_g, const ActPre& a,
                                            const Hit& h) {
    CellChange ch;
    ch.old_val = 0; ch.occ_word = 0;
    // Written flat -- one predicate per outcome, selects for the values: the reference's nest of five ifs (:312-332)
    // compiles into five levels of exec-mask bookkeeping that nearly every wavefront walks through, because one of its
    // sixteen envs usually places.  add and remove are never both set here (want_sight = add != remove).
    //   place: a free cell in front of the hit (`previous`), inside the build zone, a block of the active colour left,
    //          and the agent not standing in it
    const double x = e.x, z = e.z;
    const double y = e.y - 1.0 + PAD;  // y - (PLAYER_HEIGHT - 1) + Agent.PAD
    const double bx = (double)h.px - 0.5, by = (double)h.py, bz = (double)h.pz - 0.5;
    // (& and |, not && and ||: every operand is cheap and side-effect free, and short-circuit evaluation of floating-point
    // comparisons compiles into exec-mask branches)
    const bool overlap = (bx <= x) & (x <= bx + 1.0) & (bz <= z) & (z <= bz + 1.0) &
                         (((by <= y) & (y <= by + 1.0)) | ((by <= (y + 1.0)) & ((y + 1.0) <= by + 1.0)));
    const bool place = a.want_sight & a.add & h.hit & h.have_prev & (inv_get(e, e.active - 1) > 0) &
                       build_zone_i(h.px, h.py, h.pz) & !overlap;
    //   break: the block that was hit, unless it is the ground (GREY / WHITE cannot be broken, :330)
    const bool brk = a.want_sight & a.remove & h.hit & (h.by != -2);
    const int cx = place ? h.px : h.bx, cy = place ? h.py : h.by, cz = place ? h.pz : h.bz;
    const int cell = cell_of(cx, cy, cz);
    ch.idx = (place || brk) ? cell : -1;
    ch.lvl = cy + 1;
    ch.xz = (cx + 5) | ((cz + 5) << 4);
    ch.bit = occ_bit_hbm(cx, cy, cz);
    ch.new_val = place ? e.active : 0;   // (`previous` is never occupied: old_val = 0 for a placement)
    inv_add(e, e.active - 1, place ? -1 : 0);
    {
        // colour of the block: the one int8 the physics ever needs (L1-bypassing load: a fused rollout may have written
        // this row earlier in the same launch).  Under a WAVE-UNIFORM branch, by every lane (a lane that does not break
        // reads the first byte of its own row), into a register that holds any value otherwise: no lane-masked merge at
        // the end of the branch, so no copy of the loaded value there and no wait for it -- as a masked load merged with a
        // zero the compiler put s_waitcnt vmcnt(0) eleven instructions behind it: a memory round trip in the middle of
        // the step for every wavefront with a break, one in five.  old_colour() reads it only for a break.  (An
        // untracked inline-asm load, waited for behind the physics, is NOT safe: the compiler may copy the destination
        // register while the load is in flight, and with one lane per env it did.)
        ch.old_val = any_value();
        if (__any(brk)) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(grid_g) + (brk ? cell : 0);
            // (the atomic byte load is legalised as a 16-bit load + v_and, which again needs the value at once: the step
            // kernel, whose rows were written by earlier launches, reads with a plain load)
            if constexpr (FUSED) ch.old_val = (int)__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ch.old_val = (int)gload(src);
        }
        if (ch.idx >= 0) {
            wave_sync();
            if (G.gl == 0) {
                const int li = ch.bit + 32 * OCC_VAR0;  // the HBM row sits behind the constant prefix in LDS
                const uint32_t bit = 1u << (li & 31);
                uint32_t w = occ_s[li >> 5];
                w = ch.new_val ? (w | bit) : (w & ~bit);
                occ_s[li >> 5] = w;
                ch.occ_word = w;
            }
            wave_sync();
        }
    }
    return ch;
}

// ... and with the march on the env's own lanes (hit_test splits its work over the lanes of the group)
template <int GS, int MODE, bool PRIO = false>
__device__ inline CellChange world_act(const Grp<GS>& G, const KParams& p, Env& e, uint32_t* occ_s,
                                       const int8_t* grid_g, const TrigCtx& trig, double s0, double s1, double dy,
                                       int inventory, double cam0, double cam1, bool remove, bool add, Motion& mv,
                                       bool boost = false, uint32_t* scratch = nullptr) {
    const ActPre a = world_act_pre<GS, MODE>(G, p, e, trig, s0, s1, dy, inventory, cam0, cam1, remove, add, mv);
    Hit h;
    h.hit = false; h.have_prev = false;
    h.bx = h.by = h.bz = h.px = h.py = h.pz = 0;
    if (a.want_sight) h = hit_test<GS, PRIO>(G, occ_s, e.x, e.y, e.z, a.vx, a.vy, a.vz, boost, scratch);
    return world_act_post<GS>(G, e, occ_s, grid_g, a, h);
}

// remove_block's inventory refund (env.py:146-153 via the on_remove callback), once the colour has arrived
__device__ inline void finish_break(Env& e, CellChange& ch) {
    if (ch.idx >= 0 && ch.new_val == 0) {
        const int texture = old_colour(ch);
        if (texture >= 1 && texture <= 6) inv_add(e, texture - 1, 1);
    }
}

// World.step second half: update(dt = 1/20) (core/world.py:203-262) and the yaw wrap (:451-456)
template <int GS, int MODE, bool PRIO = false>
__device__ inline void world_update(const Grp<GS>& G, const KParams& p, Env& e, const uint32_t* occ_s,
                                    const Motion& mv, bool boost = false) {
    constexpr bool FLY = MODE == MODE_FLY;
    const int m = tis_steps(e.tis_code);
    const double dt = tis_dt(e.tis_code);   // 0.05 / m
    [[maybe_unused]] double vy_pre = 0.0;
    // A walker that starts the step well inside the padded zone cannot leave it within the step: a step moves it by
    // at most 0.25 horizontally (speed 5 x dt 0.05) and 2.5 down / 0.35 up (terminal velocity 50, jump speed 6.93),
    // and a collision push only moves it towards the centre of its cell.  When that holds for every env of the
    // wavefront -- agents return to the centre every episode -- the sub-steps skip the four comparisons of
    // build_zone (their outcome is known: inside) and the selects behind them.
    bool inside = false;
    if constexpr (!FLY) inside = __all((__builtin_fabs(e.x) < 6.0) & (__builtin_fabs(e.z) < 6.0) & (e.y >= -0.5) & (e.y < 9.0));
    bool owned = false;
    if constexpr (GS >= 4) {
        if (!IGW_DIAG_FLAG(p, 4)) {   // groups of four or more lanes: see substeps_owned
            if (!FLY && inside) vy_pre = substeps_owned<GS, FLY, true>(G, e, occ_s, mv.x, mv.y, mv.z, m, dt);   // (wave-uniform)
            else vy_pre = substeps_owned<GS, FLY, false>(G, e, occ_s, mv.x, mv.y, mv.z, m, dt);
            prio_at<PRIO, 4>(boost);
            owned = true;
        }
    }
    for (int i = 0; i < ((IGW_DIAG_FLAG(p, 4) || owned) ? 0 : m); i++) {  // _update, :222-262
        if (i == 1) prio_at<PRIO, 4>(boost);
        const double speed = FLY ? FLYING_SPEED : WALKING_SPEED;
        const double d = dt * speed;
        const double ddx = mv.x * d, ddz = mv.z * d;
        double ddy = mv.y * d;
        if (!FLY) {
            e.vy -= dt * GRAVITY;
            vy_pre = e.vy;  // time_int_steps (:243-250) is overwritten by every sub-step: only the last one's counts
            e.vy = e.vy > -TERMINAL_VELOCITY ? e.vy : -TERMINAL_VELOCITY;
        }
        ddy += e.vy * dt;
        double cx = e.x + ddx, cy = e.y + ddy, cz = e.z + ddz;
        const bool in_zone = inside || build_zone_d(cx, cy, cz, 2.0);
        if (in_zone || !FLY) {
            if (!in_zone) { cx = e.x; cz = e.z; }  // outside the padded zone a walker only moves vertically
            if constexpr (GS >= 4) collide_split<GS>(G, e, occ_s, cx, cy, cz);
            else collide(e, occ_s, cx, cy, cz);
            e.x = cx; e.y = cy; e.z = cz;
        }
    }
    if (!FLY && !IGW_DIAG_FLAG(p, 4)) e.tis_code = (vy_pre < -5.0 ? 1 : 0) + (vy_pre < -10.0 ? 1 : 0) + (vy_pre < -14.0 ? 1 : 0);   // 2 / 4 / 8 / 12 sub-steps, :243-250
    if (FLY) e.vy = 0.0;
    // yaw wrap with strict comparisons (0 and 360 both survive), :451-456
    while (e.yaw > 360.0) e.yaw -= 360.0;
    while (e.yaw < 0.0) e.yaw += 360.0;
}

// A camera delta the step accepts: finite and |v| <= IGW_CAMERA_MAX (the bound of init_pose's yaw / pitch).  The
// reference wraps yaw with `while yaw > 360: yaw -= 360` (core/world.py:451-456): a finite but huge delta (1e20:
// yaw - 360 == yaw) would spin that loop forever, on the device a hung kernel.  Anything else runs as a no-op
// component and is counted (IGW_STAT_BAD_ACTION), like the non-finite values.
__device__ inline bool camera_ok(double v) { return __builtin_fabs(v) <= IGW_CAMERA_MAX; }  // false for NaN / inf too

// parse_walking_discrete_action (core/world.py:360-394)
struct WalkAct {
    double s0, s1, dy, cam0, cam1;
    int inventory;
    bool remove, add;
};
__device__ inline WalkAct parse_walking_discrete(int action) {
    // 1 fwd (s0 = -1), 2 back (+1), 3 left (s1 = -1), 4 right (+1), 5 jump, 6..11 hotbar 1..6, 12 / 13 yaw -+ 5,
    // 14 / 15 pitch -+ 5, 16 break, 17 place; anything else is a no-op.  Written as selects: the 16 envs of a
    // wave disagree on the action, so an if-chain only buys exec-mask bookkeeping.
    const int a = action;
    const int twice = 2 * a;
    const auto pair = [&](int lo, int centre) { return (unsigned)(a - lo) < 2u ? twice - centre : 0; };
    WalkAct w;
    w.s0 = (double)pair(1, 3);             // 1 -> -1, 2 -> +1
    w.s1 = (double)pair(3, 7);             // 3 -> -1, 4 -> +1
    w.cam0 = (double)__mul24(5, pair(12, 25));   // 12 -> -5, 13 -> +5  (v_mul_i32_i24: full rate; v_mul_lo_u32 is quarter rate)
    w.cam1 = (double)__mul24(5, pair(14, 29));   // 14 -> -5, 15 -> +5
    w.dy = a == 5 ? 1.0 : 0.0;
    w.inventory = (unsigned)(a - 6) < 6u ? a - 5 : 0;
    w.remove = a == 16;
    w.add = a == 17;
    return w;
}

// Task.step_intersection (tasks/task.py:103-119) part 1: block-count delta of the synthetic grid
// (grid - start).  Only one cell can change per step, so the count is updated incrementally.
__device__ inline int syn_size_delta(const CellChange& ch, int start_val) {
    if (ch.idx < 0) return 0;
    return ((ch.new_val - start_val) != 0 ? 1 : 0) - ((old_colour(ch) - start_val) != 0 ? 1 : 0);
}

// part 2 + GridWorld.step tail (env.py:290-296) + SizeReward.step (env.py:325-331)
// The kernel parameters the END of a step needs on its common path, read from the kernarg segment in ONE batch and
// EARLY (before the physics, so that the scalar loads complete in its shadow): read where they are used they cost the
// tail of every wavefront two or three scalar-memory round trips one after the other.
struct TailParams {
    int max_steps, size_reward, autoreset;
    double right_scale, wrong_scale;
    OutRec* out;
    AgentRec* agent;
    AuxRec* aux;
    unsigned long long* stats;
    uint32_t* occ;
};
__device__ inline void tail_params_pin(TailParams& t) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(t.max_steps), "+s"(t.size_reward), "+s"(t.autoreset), "+s"(t.right_scale), "+s"(t.wrong_scale),
                 "+s"(t.out), "+s"(t.agent), "+s"(t.aux), "+s"(t.stats), "+s"(t.occ));
#endif
}
__device__ inline TailParams tail_params(const KParams& p) {
    TailParams t = {p.max_steps, p.size_reward, p.autoreset, p.right_scale, p.wrong_scale, p.out, p.agent, p.aux, p.stats, p.occ};
    tail_params_pin(t);
    return t;
}

__device__ inline StepOut finish_step(const TailParams& p, Env& e, int env_max_int, int size_new, int mi) {
    const int wrong = e.prev_size - size_new;
    bool done = mi == e.target_size;
    e.prev_size = size_new;
    const int right = mi - e.max_int;
    e.max_int = mi;
    done = done || (e.step_no == p.max_steps);
    const double reward_right = (double)right * p.right_scale, reward_wrong = (double)wrong * p.wrong_scale;
    double reward = right == 0 ? reward_wrong : reward_right;
    if (p.size_reward) {
        const int mx = max(env_max_int, e.size);
        reward = (double)(mx - e.size);
        e.size = mx;
    }
    StepOut o;
    o.reward = reward;
    o.done = done;
    return o;
}

// ---------------------------------------------------------------- incremental maximal_intersection
// tasks/task.py:147-161 as a persistent vote histogram (one 1 KB row per env: 4 rotations x 11 x 11
// admissible translations, 16-bit counts).  A step changes at most one cell, so only the target cells on
// that cell's y level whose colour equals the cell's old or new synthetic colour lose or gain a vote.  Those
// cells are not searched for: Task.__init__ on the device (prepare_tasks_kernel, the RandomTasks generator)
// leaves a COLOUR INDEX of the synthetic target next to it -- per y level the target cells sorted by colour
// (a counting sort), with the start offset of every colour class and the four rotation bounding boxes in a
// 32-byte header (IGW_LEVEL_INDEX_BYTES = 160 per level).  Per changed env the wave needs its 1 KB histogram
// row and the 160-byte index block of the changed cell's level.  Both are fetched by LDS-DMA
// (global_load_lds: no registers, no staging instructions) as soon as the change is known -- BEFORE the
// physics sub-steps -- so the loads are in flight while the wave computes.  The update itself touches LDS
// only and has NO per-env serial code: up to four changed envs are handled side by side, sixteen lanes each
// (lane = env slot | match slot | rotation): every lane takes one (matching target cell, rotation) pair
// straight from the index and votes with an LDS atomic on the staged row; the new maximum is a sixteen-lane DPP
// row reduction over the row (always exact: no rescan path, no decrement bookkeeping); the row goes back to
// HBM from the registers of the lanes that scanned it.  A wave with one changed env and a wave with four pay
// the same (the waves with many changes used to set the tail of the launch: +1,000 cycles per changed env).

// changed envs handled together per pass (one 1 KB LDS row each).  Groups of 4+ lanes rarely see more than
// four changes in a wave; narrow groups pack many envs per wave and pay LDS for their occupancy rows.
template <int GS>
constexpr int req_chunk() { return GS >= 4 ? 4 : GS == 2 ? 2 : 1; }
constexpr int LVL_BYTES = IGW_LEVEL_INDEX_BYTES;  // one level block of a task's colour index
constexpr int LVL_WORDS = LVL_BYTES / 4;
constexpr int LVL_OFFS = 16;    // byte offset of offs[N_CLASSES + 1] (bytes 0..15: the four rotation bounding boxes)
constexpr int LVL_CELLS = 32;   // byte offset of cells[<= 121]: (x << 4 | z) of the level's target cells, by class
constexpr int N_CLASSES = 14;   // synthetic colours -7..-1, 1..7 (grid - start with ids 0..7)
static_assert(LVL_CELLS + LEVEL <= LVL_BYTES && LVL_OFFS + N_CLASSES + 1 <= LVL_CELLS && LVL_BYTES % 16 == 0, "level index block layout");
static_assert(IGW_TASK_INDEX_BYTES == IGW_GRID_Y * LVL_BYTES, "task colour index layout");

// class of a synthetic colour (-1: nothing can match it: 0, or outside -7..7)
__device__ inline int colour_class(int c) { return index_class(c); }
// ... + 1 for the histogram update, which reads offs[k - 1] and offs[k]: 0 = no class; a colour outside -7..7 (block ids
// are sanitised where tasks and states enter: cannot happen) clamps to 0 / 15, and offs[15] = offs[14] is an empty slice
__device__ inline uint32_t change_class(int c) {
    const int k = min(max(c + 7 + (int)((uint32_t)c >> 31), 0), 15);   // (one v_med3_i32)
    return c == 0 ? 0u : (uint32_t)k;
}

template <int R>
struct WaveScratch {
    alignas(16) uint32_t hist[R][HIST_ROW / 2];
    alignas(16) uint32_t aux[R][LVL_WORDS];  // the colour-index block of each staged env's changed level
    alignas(8) uint32_t chg[WAVE / R][2];    // the wave's changed envs in lane order: {env, x | z << 4 | (old class + 1) << 8 | (new class + 1) << 12}
};

template <int GS>
struct BlockShared;
// (the ray march of four-lane groups parks WAVE x 10 keys in the wave's histogram rows, which are idle then)
static_assert(req_chunk<4>() * (HIST_ROW / 2) >= WAVE * 10, "hit_test scratch");
// (the RandomTasks generator builds one STRIDE-byte target row in the wave's whole scratch struct, which is idle
// at the end of a step: resolve_resets is handed &ws[wave], not one of its members)
static_assert(sizeof(WaveScratch<1>) >= (size_t)STRIDE + LVL_BYTES && alignof(WaveScratch<1>) >= 16, "sample_random_task_wave scratch");
template <int GS>
struct BlockShared {
    static constexpr int EPB = BLOCK / GS;  // envs per block
    static constexpr int EPW = WAVE / GS;   // envs per wave
    alignas(16) uint32_t occ[EPB * OCC_PITCH];  // occupancy rows, one per env
    WaveScratch<req_chunk<GS>()> ws[WAVES_PER_BLOCK];
};

__constant__ double IGW_TRIG_LUT_DEV[IGW_LUT_N * 2];

// The 2 KB table is read straight from constant memory (L2-resident, shared by every block); staging it in
// LDS cost a block-wide barrier on every launch's critical path for a handful of lookups per step.
__device__ inline const double* trig_lut() { return IGW_TRIG_LUT_DEV; }

// An env's bitmap row HBM -> LDS by the lanes of its group (lane j moves the 16-byte chunks j, j + GS, ...: a
// group reads 16 * GS contiguous bytes per instruction), plus the constant words, in two halves: occ_issue only
// issues the global loads (so the caller can queue every other load of the step behind them and pay ONE
// memory round trip), occ_commit writes LDS.
template <int GS>
struct OccStage {
    static constexpr int CH = OCC_WORDS / 4;              // 12 chunks of 16 B per env
    static constexpr int ITER = (CH + GS - 1) / GS;       // 3 at four lanes per env
    uint4 v[ITER];
};
template <int GS>
__device__ inline void occ_issue(const uint32_t* occ, const Grp<GS>& G, int env, OccStage<GS>& st) {
    const uint4* src = reinterpret_cast<const uint4*>(occ + (size_t)env * OCC_WORDS);
#pragma unroll
    for (int i = 0; i < OccStage<GS>::ITER; i++) {
        const int c = G.gl + i * GS;
        st.v[i] = (OccStage<GS>::CH % GS == 0 || c < OccStage<GS>::CH) ? src[c] : make_uint4(0, 0, 0, 0);
    }
}
template <int GS>
__device__ inline void occ_commit_var(const Grp<GS>& G, const OccStage<GS>& st, uint32_t* occ_s) {
    constexpr int CH = OccStage<GS>::CH;
#pragma unroll
    for (int i = 0; i < OccStage<GS>::ITER; i++) {
        const int c = G.gl + i * GS;
        if (CH % GS == 0 || c < CH) *reinterpret_cast<uint4*>(occ_s + OCC_VAR0 + 4 * c) = st.v[i];
    }
}
// ... the constant words of the wave's rows do not depend on any load: written while the input burst is in flight
template <int GS>
__device__ inline void occ_commit_const(uint32_t* occ_wave_s) {
    constexpr int EPW = WAVE / GS;
    // constant words: 4 lanes per env, lane part q writes the 16-byte pieces q of the prefix (words 4q..4q+3)
    // and the zero words behind the variable part
    const int lane = __lane_id();
    const int q = lane & 3;
    // words 0..9 are zero, word 10 holds the first bits of the ground plane, words 11..15 are all ones: as masks of q
    // (a four-way select of four-word constants compiles into exec-mask branches)
    static_assert(occ_const_word(0) == 0u && occ_const_word(9) == 0u && occ_const_word(10) == 0xff800000u &&
                  occ_const_word(11) == 0xffffffffu && occ_const_word(15) == 0xffffffffu, "constant prefix of the LDS occupancy row");
    const uint32_t is3 = 0u - (uint32_t)(q == 3), ge2 = 0u - (uint32_t)(q >= 2);
    const uint4 pre = make_uint4(is3, is3, is3 | (ge2 & occ_const_word(10)), ge2);
    static_assert(OCC_VAR0 == 16 && OCC_PITCH - OCC_VAR0 - OCC_WORDS == 8, "constant words are written as 4 + 2 pieces of 16 bytes");
#pragma unroll
    for (int i0 = 0; i0 < EPW; i0 += WAVE / 4) {
        const int i = i0 + (lane >> 2);
        if (EPW >= WAVE / 4 || i < EPW) {
            uint32_t* row = occ_wave_s + i * OCC_PITCH;
            *reinterpret_cast<uint4*>(row + 4 * q) = pre;
            if (q < 2) *reinterpret_cast<uint4*>(row + OCC_VAR0 + OCC_WORDS + 4 * q) = make_uint4(0, 0, 0, 0);
        }
    }
}
template <int GS>
__device__ inline void occ_commit(const Grp<GS>& G, const OccStage<GS>& st, uint32_t* occ_s, uint32_t* occ_wave_s) {
    occ_commit_var<GS>(G, st, occ_s);
    occ_commit_const<GS>(occ_wave_s);
}

// the raw action of one env, loaded before anything waits (parsed later)
struct RawAct {
    int32_t action;             // walking Discrete(18)
    uint2 buttons;              // walking Dict
    float f[5];                 // flying movement[3] + camera[2]; walking Dict camera in f[3], f[4]
    int32_t inventory, placement;
    uint32_t w1, w2;            // flying, groups of 4+ lanes: the seven dwords spread over the lanes of a quad
};
// Flying action of an env in groups of 4+ lanes: two loads instead of seven -- lane q of a quad fetches movement[q]
// (q < 3) or camera[0], then camera[1] / inventory / placement; fly_fields() hands them round by DPP at the use.
template <int GS>
__device__ inline void load_fly_spread(const ActIn& a, int env, int gl, RawAct& r) {
    const int q = gl & 3;
    // The lane's two addresses as selects on 32-bit halves and a byte offset (a select between 64-bit pointers compiles
    // into exec-mask branches): lane q reads movement[q] (q < 3) or camera[0], then camera[1] / inventory / placement.
    const auto pick = [](bool c, const void* x, const void* y) {
        const uint64_t ux = (uint64_t)(uintptr_t)x, uy = (uint64_t)(uintptr_t)y;
        const uint32_t lo = c ? (uint32_t)ux : (uint32_t)uy, hi = c ? (uint32_t)(ux >> 32) : (uint32_t)(uy >> 32);
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    const uint32_t e4 = 4u * (uint32_t)env;
    const char* b1 = pick(q < 3, a.movement, a.camera);
    const uint32_t o1 = q < 3 ? 3u * e4 + 4u * (uint32_t)q : 2u * e4;
    const char* b2 = pick(q == 0, a.camera, pick(q == 1, a.inventory, a.placement));
    const uint32_t o2 = q == 0 ? 2u * e4 + 4u : e4;
    r.w1 = *reinterpret_cast<const uint32_t*>(b1 + o1);
    r.w2 = *reinterpret_cast<const uint32_t*>(b2 + o2);
}
__device__ inline void fly_fields(RawAct& r) {
    r.f[0] = __uint_as_float((uint32_t)dpp_quad<QUAD_BCAST0>((int)r.w1));
    r.f[1] = __uint_as_float((uint32_t)dpp_quad<QUAD_BCAST1>((int)r.w1));
    r.f[2] = __uint_as_float((uint32_t)dpp_quad<QUAD_BCAST2>((int)r.w1));
    r.f[3] = __uint_as_float((uint32_t)dpp_quad<QUAD_BCAST3>((int)r.w1));
    r.f[4] = __uint_as_float((uint32_t)dpp_quad<QUAD_BCAST0>((int)r.w2));
    r.inventory = dpp_quad<QUAD_BCAST1>((int)r.w2);
    r.placement = dpp_quad<QUAD_BCAST2>((int)r.w2);
}
template <int MODE>
__device__ inline RawAct load_action(const ActIn& a, int env) {
    RawAct r = {};
    if (MODE == MODE_WALK) {
        r.action = a.actions[env];
    } else if (MODE == MODE_WALK_DICT) {
        r.buttons = *reinterpret_cast<const uint2*>(a.buttons + 8 * (size_t)env);
        r.f[3] = a.camera[2 * (size_t)env]; r.f[4] = a.camera[2 * (size_t)env + 1];
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) r.f[i] = a.movement[3 * (size_t)env + i];
        r.f[3] = a.camera[2 * (size_t)env]; r.f[4] = a.camera[2 * (size_t)env + 1];
        r.inventory = a.inventory[env];
        r.placement = a.placement[env];
    }
    return r;
}

// LDS-DMA destination: the builtin only sets M0 (the wave-uniform LDS base) when it is handed a pointer that
// is in the LDS address space by TYPE; through a generic pointer it silently emits the load without it.
// (hipcc also parses device code in its host pass, where the gfx950 builtin does not exist: hence the guard.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
#define IGW_LDS(p) ((lds_u32*)(uintptr_t)(uint32_t) reinterpret_cast<uintptr_t>(p))  /* low half of a flat LDS address = LDS offset */
__device__ inline void glds16(const void* src, uint32_t* dst) { __builtin_amdgcn_global_load_lds(src, IGW_LDS(dst), 16, 0, 0); }
__device__ inline void glds16_sc1(const void* src, uint32_t* dst) { __builtin_amdgcn_global_load_lds(src, IGW_LDS(dst), 16, 0, 16); }
// The same LDS-DMA as inline assembly -- UNTRACKED by the compiler's s_waitcnt insertion.  A tracked LDS-DMA makes the
// compiler put s_waitcnt vmcnt(0) in front of every later LDS access that might alias its destination, until it sees a
// wait of its own: the explicit asm waits below do not count, so the histogram update opened with FOUR vmcnt(0) -- each of
// which also waits for the acknowledgement of the agent-record store issued just before (the counter is in order over
// loads and stores), a store round trip at the start of the latency-bound phase of two wavefronts in three.  Untracked,
// the only waits are the explicit ones (every consumer of DMA'd rows has one: step_kernel in front of the agent store,
// resolve_changes for the later chunks).  The fused rollout keeps the builtin.  An untracked load can only make the compiler's
// own waits stricter than necessary (it counts fewer loads in flight than there are), never weaker.
// ubase: wave-uniform global address; voff: this lane's byte offset; dst: wave-uniform LDS destination (the hardware adds
// 16 x lane and the instruction offset to BOTH addresses).  M0 (the LDS base) is saved and restored.
__device__ inline void glds16u(const void* ubase, uint32_t voff, uint32_t* dst) {
    // (low half of a flat LDS address = LDS offset; readfirstlane: the operands are wave-uniform, the compiler is told so
    // -- on values that already are scalar it folds away)
    const uint32_t lds_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t) reinterpret_cast<uintptr_t>(dst));
    const uint64_t ub = (uint64_t) reinterpret_cast<uintptr_t>(ubase);
    ubase = reinterpret_cast<const void*>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ub >> 32)) << 32) |
                                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub)));
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_keep) : "s"(lds_at), "v"(voff), "s"(ubase) : "memory");
}
#else
__device__ inline void glds16(const void*, uint32_t*) {}
__device__ inline void glds16_sc1(const void*, uint32_t*) {}
__device__ inline void glds16u(const void*, uint32_t, uint32_t*) {}
#endif
// IGW_UNTRACKED_DMA: 1 = the step kernel's LDS-DMA through glds16u (see there); 0 = the compiler's builtin (rounds 3-5).
// Same-box A/B (profiles/r06_ab_variants.txt): -1.7 % walking, -0.6 % CDM, -2.6 % flying.
#ifndef IGW_UNTRACKED_DMA
#define IGW_UNTRACKED_DMA 1
#endif

// The LDS-DMA loads of one changed env into scratch slot k (whole wave): its histogram row and the colour-index
// block of the changed cell's level.  L2: bypass this CU's L1 (the fused rollout re-reads rows it stored earlier in
// the same launch, and may have regenerated the task row).
template <bool L2>
__device__ inline void dma_change_rows(const KParams& p, uint32_t* hist_dst, uint32_t* aux_dst, int env, int task, int level) {
    const int lane = __lane_id();
#if IGW_UNTRACKED_DMA
    if constexpr (!L2) {   // (the step kernel; the fused rollout keeps the builtin: its register budget has no slack, and every
                           // step of it ends with a wait for everything anyway)
        glds16u(p.hist + (size_t)env * HIST_ROW, 16u * (uint32_t)lane, hist_dst);
        if (lane < LVL_BYTES / 16) glds16u(p.task_index + (size_t)task * IGW_TASK_INDEX_BYTES + level * LVL_BYTES, 16u * (uint32_t)lane, aux_dst);
        return;
    }
#endif
    {
    const char* hrow = reinterpret_cast<const char*>(p.hist + (size_t)env * HIST_ROW) + 16 * lane;
    if (L2) glds16_sc1(hrow, hist_dst);  // cache policy sc1
    else glds16(hrow, hist_dst);
    const uint8_t* blk = p.task_index + (size_t)task * IGW_TASK_INDEX_BYTES + level * LVL_BYTES + 16 * lane;
    if (lane < LVL_BYTES / 16) {
        if (L2) glds16_sc1(blk, aux_dst);
        else glds16(blk, aux_dst);
    }
    }
}
template <int R, bool L2>
__device__ inline void dma_change_inputs(const KParams& p, WaveScratch<R>& ws, int k, int env, int task, int level) {
    dma_change_rows<L2>(p, ws.hist[k], ws.aux[k], env, task, level);
}

// The wave's changed envs are named by their leader lanes (a ballot); for the DMA their parameters travel as
// wave-uniform values (v_readlane of the leader), not through LDS.
__device__ inline int next_leader(uint64_t& m) {
    const int l = __builtin_ctzll(m);
    m &= m - 1;
    return l;
}

// Starts the DMA for the first chunk of changed envs.  Returns the ballot of leaders.
template <int GS, bool L2>
__device__ inline uint64_t prefetch_changes(const Grp<GS>& G, const KParams& p, WaveScratch<req_chunk<GS>()>& ws,
                                            bool changed, int env, int task, const CellChange& ch) {
    constexpr int R = req_chunk<GS>();
    const uint64_t mask = __ballot(changed && G.gl == 0);
    uint64_t m = mask;
#pragma unroll
    for (int k = 0; k < R; k++) {
        if (m) {
            const int l = next_leader(m);
            dma_change_inputs<R, L2>(p, ws, k, __builtin_amdgcn_readlane(env, l), __builtin_amdgcn_readlane(task, l),
                                     __builtin_amdgcn_readlane(ch.lvl, l));
        }
    }
    return mask;
}

// maximum of NON-NEGATIVE ints over each DPP row (16 lanes); valid in lane 15 of the row
__device__ inline int row_max_nonneg(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false));  // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false));  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false));  // row_shr:8
    return v;
}

// Applies the changes (after the DMA landed) and returns, per changed env, the new maximum of its histogram.
// start_val: the starting grid's value at the changed cell (synthetic colour = grid - start, env.py:290).
// Everything is lane-parallel over the up to R envs of a chunk (see the section comment): env slot = lane / 16,
// and within the sixteen lanes of a slot match slot = (lane / 4) % 4, rotation = lane % 4.
template <int GS, bool L2>
__device__ inline int resolve_changes(const Grp<GS>& G, const KParams& p, WaveScratch<req_chunk<GS>()>& ws,
                                      uint64_t mask, int env, int task, const CellChange& ch, int start_val,
                                      uint32_t* spare = nullptr) {
    constexpr int R = req_chunk<GS>();
    if (mask == 0) return 0;
    // (the fused rollout arrives with its loads in flight: the break's colour, the start byte, the DMA)
    if (L2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int lane = __lane_id();
    const int E = __builtin_popcountll(mask);
    // More changed envs than scratch rows (one wavefront in 150 to 400): the rows of the FIRST env of the second chunk
    // start moving now, into `spare` -- LDS the caller no longer needs (the step kernel: the wavefront's occupancy
    // rows, dead once the physics is done) -- and land while the first chunk is worked on.  Without it the second
    // chunk's memory round trip starts only when the first chunk's rows are free again, and such a wavefront is the
    // last one of the launch more often than not.
    uint64_t m = mask;
    if (E > R) {
#pragma unroll
        for (int k = 0; k < R; k++) m &= m - 1;   // (the first chunk's leaders: their DMA was the caller's)
        if (spare != nullptr) {
            const int l = __builtin_ctzll(m);
            dma_change_rows<L2>(p, spare, spare + HIST_ROW / 2, __builtin_amdgcn_readlane(env, l), __builtin_amdgcn_readlane(task, l),
                                __builtin_amdgcn_readlane(ch.lvl, l));
        }
    }
    // this lane's env among the changed envs of the wave: leaders below it (its own leader's bit excluded)
    const bool grp_changed = (mask >> (lane & ~(GS - 1))) & 1ull;
    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int rank = below - ((G.gl > 0 && grp_changed) ? 1 : 0);
    if (grp_changed && G.gl == 0) {
        // old / new synthetic colour (a != b) as its class + 1 -- 0: nothing in a target can match it
        const int a = old_colour(ch) - start_val, b = ch.new_val - start_val;
        *reinterpret_cast<uint2*>(ws.chg[rank]) =
            make_uint2((uint32_t)env, (uint32_t)ch.xz | (change_class(a) << 8) | (change_class(b) << 12));
    }
    const int slot = lane >> 4, sub = lane & 15, mi = sub >> 2, q = lane & 3;
    int result = 0;
    // One pass over up to R changed envs, lane-parallel (see the section comment): env slot = lane / 16, and within the
    // sixteen lanes of a slot match slot = (lane / 4) % 4, rotation = lane % 4.  `later`: a pass after the first, whose
    // slot 0 may be the spare rows.
    const auto pass = [&](int base, auto later) {
        const int cnt = min(R, E - base);
        wave_sync();
        const bool my = slot < cnt;
        const int sl = my ? slot : 0;
        const uint2 ent = my ? *reinterpret_cast<const uint2*>(ws.chg[base + sl]) : make_uint2(0u, 0u);
        uint32_t* hrow = ws.hist[sl];
        const uint8_t* blk = reinterpret_cast<const uint8_t*>(ws.aux[sl]);
        if constexpr (decltype(later)::value) {
            if (base == R && spare != nullptr && sl == 0) {
                hrow = spare;
                blk = reinterpret_cast<const uint8_t*>(spare + HIST_ROW / 2);
            }
        }
        const int gx = (int)(ent.y & 15u), gz = (int)((ent.y >> 4) & 15u);
        const int ka = (int)((ent.y >> 8) & 15u), kb = (int)((ent.y >> 12) & 15u);
        // the two colour classes' slices of the level's cell list (class + 1 == 0: an empty slice.  The reads are
        // unconditional -- a lane without an env reads slot 0's block -- and the select comes after them: no branch)
        const int oa0 = blk[LVL_OFFS - 1 + ka], oa1 = blk[LVL_OFFS + ka], ob0 = blk[LVL_OFFS - 1 + kb], ob1 = blk[LVL_OFFS + kb];
        const int na = (my && ka != 0) ? oa1 - oa0 : 0, n = na + ((my && kb != 0) ? ob1 - ob0 : 0);
        // rotation q's bounding box: admissible translations dx in [xmax - 10, xmin], dz in [zmax - 10, zmin] (task.py:62-72)
        const int bb = my ? (int)reinterpret_cast<const uint32_t*>(blk)[q] : 0;
        const int xmin = (int)(int8_t)(bb & 0xff), dxlo = (int)(int8_t)((bb >> 8) & 0xff) - 10;
        const int zmin = (int)(int8_t)((bb >> 16) & 0xff), dzlo = (int)(int8_t)((bb >> 24) & 0xff) - 10;
        for (int i0 = 0;; i0 += 4) {
            const int idx = i0 + mi;
            const bool act = idx < n;
            if (!__any(act)) break;
            if (act) {
                const bool dec = idx < na;  // a target cell of the old colour stops matching, one of the new colour starts to
                const int cell = blk[LVL_CELLS + (dec ? oa0 + idx : ob0 + (idx - na))];
                const int tx = cell >> 4, tz = cell & 15;
                // rotation q of target cell (x, z): (x,z) -> (z, 10-x) -> (10-x, 10-z) -> (10-z, x)
                // (as two selects each on predicates of the lane's rotation: a four-way select chain compiles into branches)
                const int bx = (q & 1) ? tz : tx, bz = (q & 1) ? tx : tz;
                const int rx = q >= 2 ? 10 - bx : bx;
                const int rz = (q == 1 || q == 2) ? 10 - bz : bz;
                const int u = rx - gx - dxlo, v = rz - gz - dzlo;
                if ((unsigned)u <= (unsigned)(xmin - dxlo) && (unsigned)v <= (unsigned)(zmin - dzlo)) {  // (extents <= 10: both bounds >= 0)
                    const int bin = q * 121 + __mul24(u, 11) + v;
                    atomicAdd(&hrow[bin >> 1], (dec ? 0xffffffffu : 1u) << (16 * (bin & 1)));   // -/+ 1 in the bin's half
                }
            }
        }
        wave_sync();
        // every slot's sixteen lanes read their env's updated row once: it goes back to HBM from these registers and
        // its maximum is a DPP row reduction (the eight 16-bit counts of a piece: packed maxima, v_pk_max_u16)
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        us2 pm = {0, 0};
        if (my) {
            const uint4* row = reinterpret_cast<const uint4*>(hrow);
            uint4* dst = reinterpret_cast<uint4*>(p.hist + (size_t)ent.x * HIST_ROW);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint4 now = row[j * 16 + sub];
                dst[j * 16 + sub] = now;  // (plain stores: non-temporal ones measured no better, the CDM workload worse)
                pm = __builtin_elementwise_max(pm, __builtin_elementwise_max(
                    __builtin_elementwise_max(__builtin_bit_cast(us2, now.x), __builtin_bit_cast(us2, now.y)),
                    __builtin_elementwise_max(__builtin_bit_cast(us2, now.z), __builtin_bit_cast(us2, now.w))));
            }
        }
        // ... valid in lane 15 of the slot's sixteen lanes: the env's lanes fetch it from there (a cross-lane read
        // through the LDS crossbar: no LDS write / barrier / read)
        const int best = row_max_nonneg((int)max((uint32_t)pm.x, (uint32_t)pm.y));
        const int mine = __builtin_amdgcn_ds_bpermute((((rank - base) & (R - 1)) * 16 + 15) * 4, best);
        if (grp_changed && rank >= base && rank < base + R) result = mine;
    };
    // (the step kernel gets the first pass peeled: its common path carries none of the later chunks' bookkeeping.  The
    // fused rollout keeps ONE copy of the pass in one loop: two copies cost it three registers more than its launch bound has)
    if constexpr (!L2) pass(0, std::false_type{});
    for (int base = L2 ? 0 : R; base < E; base += R) {   // (rare beyond the first) the later chunks are fetched only now
        if (!L2 || base > 0) {
            if (IGW_DIAG_FLAG(p, 256)) break;  // diag 256: what the later passes of a wave with more than R changes cost
            const int cnt = min(R, E - base);
#pragma unroll
            for (int k = 0; k < R; k++) {
                if (k < cnt) {
                    const int l = next_leader(m);
                    if (!(k == 0 && base == R && spare != nullptr))   // (already on its way)
                        dma_change_inputs<R, L2>(p, ws, k, __builtin_amdgcn_readlane(env, l), __builtin_amdgcn_readlane(task, l),
                                                 __builtin_amdgcn_readlane(ch.lvl, l));
                }
            }
            // the first chunk's DMA was waited for by the caller (before it issued its output stores); later chunks wait here
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        pass(base, std::true_type{});
    }
    // a fused rollout reads the rows again in its next step: let the stores reach L2 first
    if (L2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    return result;
}

// auto-reset rows of every done env of this wave (whole wave per env, coalesced).  With the RandomTasks
// generator on, the env's task row is regenerated first (row_s: 1104 bytes of LDS scratch) and the lanes of
// the env learn the new target size through `generated_size`.
template <int GS, bool RT>
__device__ inline void resolve_resets(const Grp<GS>& G, const KParams& p, bool do_reset, int env, int task,
                                      bool has_start, uint32_t ep, uint32_t* occ_wave_s, int8_t* row_s,
                                      int& generated_size, int staged_leader = -1, const uint32_t* staged_row = nullptr,
                                      const uint32_t* staged_tail = nullptr, uint32_t staged_occ = 0) {
    uint64_t m = __ballot(do_reset);
    while (m) {
        const int l = __builtin_ctzll(m);
        const int gsel = l / GS;
        if constexpr (GS == 64) m = 0;
        else if constexpr (GS == 1) m &= m - 1;
        else m &= ~(((1ull << GS) - 1ull) << (gsel * GS));
        const int t_env = __builtin_amdgcn_readlane(env, l);
        const int t_task = __builtin_amdgcn_readlane(task, l);
        int t_hs = __builtin_amdgcn_readlane((int)has_start, l);
        if constexpr (RT) {
            if (p.rt_enabled) {
                const int n = sample_random_task_wave(p, t_env, (uint32_t)__builtin_amdgcn_readlane((int)ep, l), row_s);
                if (__lane_id() / GS == gsel) generated_size = n;
                t_hs = 0;
            }
        }
        if (l == staged_leader) reset_rows_staged(p, t_env, staged_row, staged_tail, staged_occ);
        else reset_rows_wave(p, t_env, t_task, t_hs != 0, occ_wave_s ? occ_wave_s + gsel * OCC_PITCH : nullptr);
    }
}

// One IGW_TRAJ_BYTES record of the episode log (include/igw.h) by the env's leader lane; `ep`: the episode the
// step belongs to.  Actions are re-read from the caller's buffers (nothing is kept in registers for the log).
template <int MODE>
__device__ inline void write_trajectory(const KParams& p, const ActIn& a, int env, int task, uint32_t ep, const Env& e,
                                        const CellChange& ch, const StepOut& o, bool was_reset, int task_next) {
    int32_t* head = p.traj_heads + ((size_t)env * 2 + (ep & 1)) * 4;
    const int step = e.step_no;  // 1-based, already counted
    if (step >= 1 && step <= p.traj_cap) {
        uint32_t rec[16];
        rec[0] = __float_as_uint((float)e.x); rec[1] = __float_as_uint((float)e.y); rec[2] = __float_as_uint((float)e.z);
        rec[3] = __float_as_uint((float)e.pitch); rec[4] = __float_as_uint((float)e.yaw);
        rec[5] = __float_as_uint((float)o.reward);
        rec[6] = __float_as_uint((float)(e.yaw - 180.0));
        rec[7] = e.inv01; rec[8] = e.inv23; rec[9] = e.inv45;
        uint32_t tag = (uint32_t)MODE;
        for (int i = 11; i < 16; i++) rec[i] = 0;
        if (MODE == MODE_WALK) {
            rec[11] = (uint32_t)a.actions[env];
        } else if (MODE == MODE_FLY) {
            for (int i = 0; i < 3; i++) rec[11 + i] = __float_as_uint(a.movement[3 * (size_t)env + i]);
            for (int i = 0; i < 2; i++) rec[14 + i] = __float_as_uint(a.camera[2 * (size_t)env + i]);
            const int inv = a.inventory[env], plc = a.placement[env];   // as executed: a rejected id runs as 0, placement 1 / 2 / other
            tag |= ((unsigned)inv > 6u ? 0u : (uint32_t)inv << 2) | ((plc == 1 ? 1u : plc == 2 ? 2u : 0u) << 5);
        } else {
            const uint2 bw = *reinterpret_cast<const uint2*>(a.buttons + 8 * (size_t)env);
            rec[11] = bw.x; rec[12] = bw.y;
            for (int i = 0; i < 2; i++) rec[13 + i] = __float_as_uint(a.camera[2 * (size_t)env + i]);
        }
        const uint32_t change = ch.idx < 0 ? 0xffffu : ((uint32_t)ch.idx | (((uint32_t)ch.new_val & 7u) << 11));
        rec[10] = change | ((o.done ? 1u : 0u) << 16) | (tag << 24);
        uint4* dst = reinterpret_cast<uint4*>(p.traj + (((size_t)env * 2 + (ep & 1)) * p.traj_cap + (step - 1)) * IGW_TRAJ_BYTES);
#pragma unroll
        for (int i = 0; i < 4; i++) dst[i] = make_uint4(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]);
    }
    *reinterpret_cast<int4*>(head) = make_int4(task, min(step, p.traj_cap), (int)ep, o.done ? 1 : 0);
    if (was_reset)  // the next episode's slot starts empty
        *reinterpret_cast<int4*>(p.traj_heads + ((size_t)env * 2 + ((ep + 1) & 1)) * 4) = make_int4(task_next, 0, (int)(ep + 1), 0);
}


// The kernel parameters as the launch's kernarg segment, behind a pointer the compiler cannot see through: code
// that reads them through it loads what it needs where it needs it (scalar loads from the constant cache) instead
// of keeping everything it will need at the END of the step in scalar registers from the START (the step kernel
// ran out of them and parked 29 in vector-register lanes: v_writelane / v_readlane are vector-ALU instructions).
constexpr int STEP_KERNARG_HEAD = 56;   // the step kernel's seven leading pointer arguments
#if defined(__HIP_DEVICE_COMPILE__)
template <int OFFSET = 0>
__device__ inline const KParams& kernarg_again(const KParams&) {
    typedef __attribute__((address_space(4))) const KParams kparams_c;
    kparams_c* ka = (kparams_c*)((__attribute__((address_space(4))) const char*)__builtin_amdgcn_kernarg_segment_ptr() + OFFSET);
    asm volatile("" : "+s"(ka));
    return *(const KParams*)ka;
}
#else
template <int OFFSET = 0>
__device__ inline const KParams& kernarg_again(const KParams& p) { return p; }
#endif

// What the in-step reset reads of the kernel parameters (without the generator / episode-log extras), as one batch of
// scalar loads: every other field of the copy is dead.
__device__ inline KParams reset_params(const KParams& p) {
    KParams q = {};
    q.sample_tasks = p.sample_tasks; q.n_tasks = p.n_tasks; q.rt_enabled = p.rt_enabled;
    q.sample_seed = p.sample_seed; q.env_base = p.env_base;
    q.task_meta = p.task_meta; q.grid = p.grid; q.occ = p.occ; q.hist = p.hist;
    q.task_start = p.task_start; q.task_start_occ = p.task_start_occ;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(q.sample_tasks), "+s"(q.n_tasks), "+s"(q.rt_enabled), "+s"(q.sample_seed), "+s"(q.env_base),
                 "+s"(q.task_meta), "+s"(q.grid), "+s"(q.occ), "+s"(q.hist), "+s"(q.task_start), "+s"(q.task_start_occ));
#endif
    return q;
}

// The end of a step: reward / done, the in-step reset of the episodes that ended, the last stores, the counters.
// Two copies of the stores, chosen by a wave-uniform branch: the fifteen wavefronts in sixteen that reset nothing must not
// run BEHIND the reset path -- where the two paths join, the compiler is conservative: an s_waitcnt vmcnt(0) for a load
// that only the reset path issues, which on the common path waits for the acknowledgement of every store of the
// wavefront, and a dozen copies that merge the reset values with the step's.  size_in: SizeReward.size as the step loaded
// it (the aux record goes back to memory only when it changed).
template <int GS, int MODE, bool EXTRA>
__device__ inline void tail_step(const Grp<GS>& G, const KParams& p, const ActIn& a, BlockShared<GS>& sh, int wave, int env,
                                 bool active, Env& e, const CellChange& ch, int task, int env_max_int, int size_new, int mi,
                                 bool need, bool changed, int8_t* grid_g, const uint32_t* occ_s, bool boost,
                                 [[maybe_unused]] int diag_m, ResetMeta rm, bool pre_ok, const TailParams& tp, int size_in,
                                 int staged_leader, uint32_t staged_occ, int block_envs) {
    const StepOut o = finish_step(tp, e, env_max_int, size_new, mi);
    const bool do_reset = active && o.done && tp.autoreset;
    // the stores of an env whose episode goes on (`keep`); the counters of the wavefront
    const auto step_stores = [&](bool keep) {
        if constexpr (EXTRA) {
            if (keep && p.traj && env < p.traj_n && G.gl == 0) write_trajectory<MODE>(p, a, env, task, e.episode, e, ch, o, false, task);
        }
        if (keep && G.gl == 0 && ch.idx >= 0) {
            gstore(grid_g + ch.idx, (int8_t)ch.new_val);
            gstore(tp.occ + (size_t)env * OCC_WORDS + (ch.bit >> 5), ch.occ_word);
        }
        const bool aux_dirty = changed || e.size != size_in;
        if constexpr (GS >= 4) {
            // the whole output record in ONE store instruction: observations (env.py:281-289) by lanes 0..2, reward + done
            // by lane 3
            if (keep && G.gl < 4 && !IGW_DIAG_FLAG(p, 16))
                st4(reinterpret_cast<uint4*>(tp.out + env) + G.gl, G.gl == 3 ? out_piece_result((float)o.reward, o.done) : out_piece_step(e, G.gl));
            if (keep && G.gl == 0 && aux_dirty) st4(tp.aux + env, env_pack_aux(e, task));
        } else if (keep && G.gl == 0) {
            if (!IGW_DIAG_FLAG(p, 16)) out_store(tp.out + env, e, false, (float)o.reward, o.done);
            if (aux_dirty) aux_store(e, task, tp.aux + env);
        }
    };
    const auto counters = [&]() {   // one branch for the wavefronts with nothing to count, scalar counts from one lane otherwise
#if IGW_EARLY_COUNTERS
        const uint64_t m_need = 0, m_cell = 0;   // (added where they became known: step_kernel)
        const uint64_t m_reset = __ballot(do_reset && G.gl == 0);
#else
        const uint64_t m_need = __ballot(need && active && G.gl == 0), m_cell = __ballot(ch.idx >= 0 && active && G.gl == 0),
                       m_reset = __ballot(do_reset && G.gl == 0);
#endif
        // IGW_STAT_STEPS: the env-steps of the launch -- the device-side count of the work a launch did (bench.py gathers
        // its per-rank delta over RCCL).  Added ONCE PER LAUNCH, by the first wavefront of block 0 (a scalar condition;
        // n_envs is read from the kernarg segment inside the branch).  Round 5 added it once per BLOCK: the first
        // wavefront of every block then always took this branch -- 1,024 atomics per 65,536-env launch, sixteen to a
        // stripe -- and the launch was 1-2 % slower (same-box A/B: profiles/r06_ab_variants.txt).
#if IGW_STEPS_MODE == 2
        const bool count_steps = wave == 0 && blockIdx.x == 0;
#elif IGW_STEPS_MODE == 1
        const bool count_steps = wave == 0;
#else
        const bool count_steps = false;
#endif
        if (((m_need | m_cell | m_reset) != 0 || count_steps) && tp.stats != nullptr && G.lane == 0) {
            unsigned long long* st = tp.stats + (blockIdx.x & (IGW_STAT_STRIPES - 1)) * 8;
            if (count_steps) counter_add(st + IGW_STAT_STEPS, (unsigned long long)(IGW_STEPS_MODE == 2 ? p.n_envs : block_envs));
            if (m_need) counter_add(st + IGW_STAT_CHANGED, (unsigned long long)__builtin_popcountll(m_need));
            if (m_cell) counter_add(st + IGW_STAT_RESCANS, (unsigned long long)__builtin_popcountll(m_cell));
            if (m_reset) counter_add(st + IGW_STAT_RESETS, (unsigned long long)__builtin_popcountll(m_reset));
        }
    };
#ifdef IGW_DIAG
    {
        const unsigned long long n_ch = __builtin_popcountll(__ballot(changed && G.gl == 0));
        const unsigned long long n_rt = __builtin_popcountll(__ballot(do_reset && G.gl == 0));
        const unsigned long long n_hit = __builtin_popcountll(__ballot(ch.idx >= 0 && ch.new_val == 0 && G.gl == 0));
        stamp_features(p, n_ch | (n_rt << 16) | ((unsigned long long)wave_max_i32(diag_m) << 24) | (n_hit << 32));
    }
#endif
    if (!__any(do_reset)) {   // fifteen wavefronts in sixteen
        step_stores(active);
        counters();
        stamp(p, 6);
        return;
    }
    // ---- a wavefront with an episode that ended: the reset first (its loads go out before the stores), then the stores
    // The kernel parameters the reset path needs, read from the kernarg segment in ONE batch: read where they are used,
    // each of eight scalar loads was followed by its own s_waitcnt -- eight scalar-memory round trips one after the other
    // in the wavefronts that already have the most to do.
    KParams rq;
    if constexpr (EXTRA) rq = p;   // (the generator and the episode log use most of them)
    else rq = reset_params(p);
    const KParams& rp = rq;
    uint32_t ep = e.episode;
    const int task_old = task;
    int generated_size = -1;
    bool has_start = false;
    ResetVals rv = {};
    if (do_reset) {
        ep = next_task(rp, env, e, task);  // the next episode's task (task generators on the device)
        // (an episode that ends early -- target completed -- or a task that was only just chosen: fetched now)
        if (!pre_ok) rm = load_reset_meta(rp.task_meta + task);
        rv = reset_decode(rm);
        has_start = !rp.rt_enabled && rv.has_start;
    }
    constexpr int RS = req_chunk<GS>() - 1;   // (the scratch slot the step kernel staged the starting row into)
    resolve_resets<GS, EXTRA>(G, rp, do_reset, env, task, has_start, ep, nullptr,
                              reinterpret_cast<int8_t*>(&sh.ws[wave]), generated_size, staged_leader, sh.ws[wave].hist[RS],
                              sh.ws[wave].aux[RS], staged_occ);
    prio_at<true, 7>(boost);
    step_stores(active && !do_reset);
    if (do_reset) {
        if constexpr (EXTRA) {
            if (p.traj && env < p.traj_n && G.gl == 0) write_trajectory<MODE>(p, a, env, task_old, ep, e, ch, o, true, task);
        }
        reset_env_regs(e, rv, false, generated_size);
        if constexpr (GS >= 4) {
            if (G.gl < 4) {   // the reset's agent record goes the way the step's went (same lane -> same address: program order)
                st4(reinterpret_cast<uint4*>(tp.agent + env) + G.gl, agent_piece(G.gl, e.x, e.y, e.z, e.yaw, e.pitch, e.vy, env_pack_piece3(e)));
                if (!IGW_DIAG_FLAG(p, 16))   // the whole output record: the reset's observations, the step's reward + done
                    st4(reinterpret_cast<uint4*>(tp.out + env) + G.gl, G.gl == 3 ? out_piece_result((float)o.reward, o.done) : out_piece_reset(e, G.gl));
            }
            if (G.gl == 0) st4(tp.aux + env, env_pack_aux(e, task));
        } else if (G.gl == 0) {
            env_store(e, tp.agent + env);
            if (!IGW_DIAG_FLAG(p, 16)) out_store(tp.out + env, e, true, (float)o.reward, o.done);
            aux_store(e, task, tp.aux + env);
        }
    }
    counters();
    stamp(p, 6);
}

// __launch_bounds__(BLOCK, 4): four waves per SIMD, i.e. at most 128 VGPRs -- the whole 65,536-env batch at four
// lanes per env is then co-resident (4,096 waves = 4 per SIMD) and runs in one round.
// EXTRA: the RandomTasks generator and the episode log are compiled in (launched only when one of them is enabled,
// so the plain kernel carries neither their code nor their registers).  Narrow groups pack so many envs per block
// that LDS (one occupancy row per env) caps them at 2-3 blocks per CU anyway.
template <int GS, int MODE, bool EXTRA, bool EXACT = false>
__global__ __launch_bounds__(BLOCK, (GS >= 4 ? 4 : 2)) void step_kernel(const uint32_t* h_occ, const AgentRec* h_agent, const AuxRec* h_aux,
                                                                        const void* h_a0, const void* h_a1, const void* h_a2, const void* h_a3,
                                                                        KParams p, ActIn a) {
    // The seven leading arguments are what the input burst needs -- occupancy rows, agent records, aux records, the
    // action buffers (walking: h_a0 = actions; Dict: buttons, camera; flying: movement, camera, inventory, placement)
    // and, where a slot is free (h_a3, walking and Dict), the number of envs.  The library is built with
    // -amdgpu-kernarg-preload-count=7: they arrive in scalar registers with the wavefront, and the first loads of the
    // step are issued without the scalar-load round trip to the kernarg segment (-0.37 us per launch, same-box A/B).
    // Flying has no free slot: EXACT = the batch is a whole number of blocks (no inactive lanes, no bound to read).
    __shared__ BlockShared<GS> sh;
    const Grp<GS> G;
    TrigCtx trig;
    trig.lut = trig_lut();
    // (wave-uniform, and the compiler is told so: the addresses of the wavefront's LDS scratch are scalar arithmetic)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int slot = threadIdx.x / GS;
    const int env = blockIdx.x * BlockShared<GS>::EPB + slot;
    const int wave_env0 = blockIdx.x * BlockShared<GS>::EPB + wave * BlockShared<GS>::EPW;
    static_assert(!EXACT || MODE == MODE_FLY, "EXACT is the flying kernel's variant");
    const int n_envs = MODE != MODE_FLY ? (int)(uintptr_t)h_a3 : EXACT ? 0x7fffffff : p.n_envs;
    const bool active = EXACT || env < n_envs;
    uint32_t* occ_wave_s = sh.occ + wave * BlockShared<GS>::EPW * OCC_PITCH;
    uint32_t* occ_s = sh.occ + slot * OCC_PITCH;
    if (!EXACT && wave_env0 >= n_envs) return;
    if (IGW_DIAG_FLAG(p, 64)) return;  // diag 64: the empty launch (same grid, registers and LDS)
    stamp(p, 0);
    prio_at<true, 0>();
    // Lanes past the last env (only in the last wave, when N is not a multiple of the envs per wave) run on a copy
    // of the last env and store nothing: the step below has no "is this lane alive" control flow.
    const int env_r = active ? env : n_envs - 1;
    const bool writer = active && G.gl == 0;
    // Every load the step needs before it can compute -- occupancy row, agent record, task index, action -- is
    // issued before the first wait: one memory round trip.
    OccStage<GS> occ_in = {};
    constexpr bool SPLIT = IGW_SPLIT_BURST && GS == 4;   // (four-lane groups: "Split burst" below; the other widths: occupancy first, one wait)
    if constexpr (!SPLIT) {
        if (!IGW_DIAG_FLAG(p, 32)) occ_issue<GS>(h_occ, G, env_r, occ_in);  // diag 32: what the occupancy rows cost in the load burst
    }
    // The agent record: with four or more lanes per env, lane q of a quad fetches the 16-byte piece q and the quad
    // hands the pieces round by DPP after the wait -- ONE dwordx4 per lane instead of four.  (All four lanes reading
    // the whole record is one cache line per env either way, but the CU's address pipe handles a wavefront's
    // dwordx4 in 16 cycles whatever the addresses, and sixteen wavefronts per CU queue up behind one another there:
    // the input burst of a wavefront was 48 + 64 + 8 of those cycles, the record more than half of it.)
    // The aux record (episode ints, task row, episode counter) likewise: lane q fetches its dword q.
    constexpr bool REC_SPREAD = GS >= 4;
    AgentRec rec;
    uint4 rec_piece = make_uint4(0, 0, 0, 0), aux = make_uint4(0, 0, 0, 0);
    uint32_t aux_w = 0;
    if constexpr (REC_SPREAD) {
        rec_piece = reinterpret_cast<const uint4*>(h_agent + env_r)[G.gl & 3];
        aux_w = reinterpret_cast<const uint32_t*>(h_aux + env_r)[G.gl & 3];
    } else {
        rec = h_agent[env_r];  // every lane of the group reads the same 64 B line (one request)
        aux = *reinterpret_cast<const uint4*>(h_aux + env_r);
    }
    constexpr bool FLY_SPREAD = MODE == MODE_FLY && GS >= 4;
    const ActIn ah = {(const int32_t*)h_a0, (const float*)h_a0, (const float*)h_a1, (const int32_t*)h_a2, (const int32_t*)h_a3,
                      (const uint8_t*)h_a0};   // (the fields of the mode's own action space are the valid ones)
    RawAct ra = {};
    if constexpr (FLY_SPREAD) load_fly_spread<GS>(ah, env_r, G.gl, ra);
    else ra = load_action<MODE>(ah, env_r);
    int8_t* grid_g = p.grid + (size_t)env_r * STRIDE;
    // Split burst (four-lane groups): the SMALL loads (agent record piece, aux word, action: 24-28 bytes per lane) are
    // issued FIRST and waited for alone -- loads return in order, so s_waitcnt vmcnt(3) -- and the record unpack, the
    // action parse and the camera / heading arithmetic run while the three occupancy pieces (48 bytes per lane, 70 % of
    // the burst) are still on their way; the occupancy words go to LDS behind world_act_pre, in front of the ray march.
    if constexpr (SPLIT) {
        if (!IGW_DIAG_FLAG(p, 32)) occ_issue<GS>(h_occ, G, env_r, occ_in);
    }
    occ_commit_const<GS>(occ_wave_s);   // (LDS writes that need no load: in the shadow of the burst)
#if defined(__HIP_DEVICE_COMPILE__) && IGW_SPLIT_BURST
    if constexpr (GS == 4) {   // the first wait: the small loads only
        if constexpr (MODE == MODE_WALK)
            asm volatile("" : "+v"(aux_w), "+v"(ra.action), "+v"(rec_piece.x), "+v"(rec_piece.y), "+v"(rec_piece.z), "+v"(rec_piece.w));
        else if constexpr (MODE == MODE_FLY)
            asm volatile("" : "+v"(aux_w), "+v"(ra.w1), "+v"(ra.w2), "+v"(rec_piece.x), "+v"(rec_piece.y), "+v"(rec_piece.z), "+v"(rec_piece.w));
        else
            asm volatile("" : "+v"(aux_w), "+v"(ra.buttons.x), "+v"(ra.buttons.y), "+v"(ra.f[3]), "+v"(ra.f[4]), "+v"(rec_piece.x), "+v"(rec_piece.y), "+v"(rec_piece.z), "+v"(rec_piece.w));
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !IGW_SPLIT_BURST
    // ... and ONE wait: the empty statement below takes a register of every load above as an operand, and the
    // occupancy words as in-out operands, so every load is issued before it and the LDS writes of the occupancy row
    // come after it.  (Without it the compiler waited for the occupancy row, wrote it to LDS, and only then issued
    // the loads of the agent record, the aux record and the action: two memory round trips, 2.6 K cycles, at the head
    // of every wavefront, in round 2 as well.)
    if constexpr (GS == 4) {
        uint32_t &o0 = occ_in.v[0].x, &o1 = occ_in.v[1].x, &o2 = occ_in.v[2].x;
        if constexpr (MODE == MODE_WALK)
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2) : "v"(aux_w), "v"(ra.action), "v"(rec_piece.x), "v"(rec_piece.y), "v"(rec_piece.z), "v"(rec_piece.w));
        else if constexpr (MODE == MODE_FLY)
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2) : "v"(aux_w), "v"(ra.w1), "v"(ra.w2), "v"(rec_piece.x), "v"(rec_piece.y), "v"(rec_piece.z), "v"(rec_piece.w));
        else
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2) : "v"(aux_w), "v"(ra.buttons.x), "v"(ra.f[3]), "v"(ra.f[4]), "v"(rec_piece.x), "v"(rec_piece.y), "v"(rec_piece.z), "v"(rec_piece.w));
    }
#endif
    if constexpr (!SPLIT) occ_commit_var<GS>(G, occ_in, occ_s);
    if constexpr (REC_SPREAD) {  // piece 0: x, y | 1: z, yaw | 2: pitch, vy | 3: inventory, step_no, pack (include/igw.h)
        uint32_t w[16];
        const int px = (int)rec_piece.x, py = (int)rec_piece.y, pz = (int)rec_piece.z, pw = (int)rec_piece.w;
        w[0] = (uint32_t)dpp_quad<QUAD_BCAST0>(px); w[1] = (uint32_t)dpp_quad<QUAD_BCAST0>(py);
        w[2] = (uint32_t)dpp_quad<QUAD_BCAST0>(pz); w[3] = (uint32_t)dpp_quad<QUAD_BCAST0>(pw);
        w[4] = (uint32_t)dpp_quad<QUAD_BCAST1>(px); w[5] = (uint32_t)dpp_quad<QUAD_BCAST1>(py);
        w[6] = (uint32_t)dpp_quad<QUAD_BCAST1>(pz); w[7] = (uint32_t)dpp_quad<QUAD_BCAST1>(pw);
        w[8] = (uint32_t)dpp_quad<QUAD_BCAST2>(px); w[9] = (uint32_t)dpp_quad<QUAD_BCAST2>(py);
        w[10] = (uint32_t)dpp_quad<QUAD_BCAST2>(pz); w[11] = (uint32_t)dpp_quad<QUAD_BCAST2>(pw);
        w[12] = (uint32_t)dpp_quad<QUAD_BCAST3>(px); w[13] = (uint32_t)dpp_quad<QUAD_BCAST3>(py);
        w[14] = (uint32_t)dpp_quad<QUAD_BCAST3>(pz); w[15] = (uint32_t)dpp_quad<QUAD_BCAST3>(pw);
        __builtin_memcpy(&rec, w, sizeof(rec));
        aux = make_uint4((uint32_t)dpp_quad<QUAD_BCAST0>((int)aux_w), (uint32_t)dpp_quad<QUAD_BCAST1>((int)aux_w),
                         (uint32_t)dpp_quad<QUAD_BCAST2>((int)aux_w), (uint32_t)dpp_quad<QUAD_BCAST3>((int)aux_w));
    }
    Env e;
    env_unpack(e, rec, aux);
    int task = (int)aux.z;
    const int size_in = e.size;
    // an episode that reaches max_steps in this step is reset inside the kernel: what the reset needs of its task's
    // metadata row is fetched now (ResetMeta; with a task generator on the next task is not known yet)
    const bool ends = p.autoreset && e.step_no + 1 >= p.max_steps;
    const bool pre_ok = ends && !p.sample_tasks && !p.rt_enabled;
    // (the lanes that do not prefetch: tail_step loads for them if it has to.  Their registers hold "any value", and the
    // cheapest any value is one that is already in registers and dead: the words of the input burst)
    ResetMeta pre = reset_meta_any();
    if constexpr (GS == 4 && REC_SPREAD && !SPLIT) {
        const auto as4 = [](const uint4& v) { return vu4{v.x, v.y, v.z, v.w}; };
        pre = ResetMeta{as4(rec_piece), as4(occ_in.v[0]), as4(occ_in.v[1]), as4(occ_in.v[2])};
    }
    if (pre_ok) pre = load_reset_meta(p.task_meta + task);
    if constexpr (!SPLIT) wave_sync();
    stamp(p, 1);
    if (IGW_DIAG_FLAG(p, 128)) return;  // diag 128: launch + the input burst, nothing else
    // a wave with an episode running out in this step has the reset to do on top: one priority level up
    const bool boost = __any(ends);
    prio_at<true, 1>(boost);
    [[maybe_unused]] const int diag_m = tis_steps(e.tis_code);  // IGW_DIAG: sub-steps this env asked for
    e.step_no = min(e.step_no + 1, 65535);  // env.py:276
    CellChange ch;
    Motion mv;
    ActPre ap;
    if (MODE == MODE_WALK) {
        const WalkAct w = parse_walking_discrete(ra.action);
        ap = world_act_pre<GS, MODE_WALK>(G, p, e, trig, w.s0, w.s1, w.dy, w.inventory, w.cam0, w.cam1, w.remove, w.add, mv);
    } else if (MODE == MODE_WALK_DICT) {  // parse_walking_action, core/world.py:396-414
        const uint2 bw = ra.buttons;
        const bool fwd = bw.x & 0xffu, back = bw.x & 0xff00u, left = bw.x & 0xff0000u, right = bw.x & 0xff000000u;
        const bool jump = bw.y & 0xffu, attack = bw.y & 0xff00u, use = bw.y & 0xff0000u;
        int hotbar = (int)(bw.y >> 24);
        double c0 = (double)ra.f[3], c1 = (double)ra.f[4];
        // the reference raises on these (core/world.py:354-355) or would carry NaN into the pose: run the
        // offending component as a no-op and count it (IGW_STAT_BAD_ACTION)
        bool bad = false;
        if (hotbar > 6) { hotbar = 0; bad = true; }
        if (!camera_ok(c0)) { c0 = 0.0; bad = true; }
        if (!camera_ok(c1)) { c1 = 0.0; bad = true; }
        if (bad && writer) stat_add(p.stats, IGW_STAT_BAD_ACTION, 1);
        const double s0 = (fwd ? -1.0 : 0.0) + (back ? 1.0 : 0.0), s1 = (left ? -1.0 : 0.0) + (right ? 1.0 : 0.0);
        ap = world_act_pre<GS, MODE_WALK_DICT>(G, p, e, trig, s0, s1, jump ? 1.0 : 0.0, hotbar, c0, c1, attack, use, mv);
    } else {  // parse_flying_action, core/world.py:416-432
        if constexpr (FLY_SPREAD) fly_fields(ra);
        const int placement = ra.placement;
        int inventory = ra.inventory;
        double f[5] = {(double)ra.f[0], (double)ra.f[1], (double)ra.f[2], (double)ra.f[3], (double)ra.f[4]};
        bool bad = false;  // see the walking Dict branch
        if ((unsigned)inventory > 6u) { inventory = 0; bad = true; }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (i < 3 ? !__builtin_isfinite(f[i]) : !camera_ok(f[i])) { f[i] = 0.0; bad = true; }
        }
        if (bad && writer) stat_add(p.stats, IGW_STAT_BAD_ACTION, 1);
        ap = world_act_pre<GS, MODE_FLY>(G, p, e, trig, f[0], f[1], f[2], inventory, f[3], f[4], placement == 2, placement == 1, mv);
    }
#ifdef IGW_DIAG
    if (IGW_DIAG_FLAG(p, 512)) {   // diag 512: input burst + action parse + world_act_pre, nothing else (results kept alive)
        asm volatile("" :: "v"(mv.x), "v"(mv.z), "v"(mv.y), "v"(ap.vx), "v"(ap.vy), "v"(ap.vz), "v"(e.yaw), "v"(e.pitch), "v"(e.vy), "v"(e.active));
        return;
    }
#endif
    if constexpr (SPLIT) {   // the occupancy pieces have had the record unpack, the action parse and the step's trig to arrive
        occ_commit_var<GS>(G, occ_in, occ_s);
        wave_sync();
    }
    // hit_test (core/world.py:73-99) for the envs that place or break -- 8 of the 18 walking actions.
    Hit h;
    h.hit = false; h.have_prev = false;
    h.bx = h.by = h.bz = h.px = h.py = h.pz = 0;
    if (ap.want_sight) h = hit_test<GS, true>(G, occ_s, e.x, e.y, e.z, ap.vx, ap.vy, ap.vz, boost, sh.ws[wave].hist[0]);
    ch = world_act_post<GS, false>(G, e, occ_s, grid_g, ap, h);
    // issued here, consumed after the histogram update
    int start_val = 0, env_max_int = 0;
    if (ch.idx >= 0) start_val = p.task_start[(size_t)task * STRIDE + ch.idx];
    if (p.size_reward && e.step_no == 1) env_max_int = p.task_meta[task].env_max_int;
    stamp(p, 2);
    const bool changed = active && ch.idx >= 0 && !IGW_DIAG_FLAG(p, 1);
    // the histogram row and the colour-index block of the changed level of every changed env start moving into LDS
    // now (LDS-DMA) and land while the physics runs
    const uint64_t chg_mask = prefetch_changes<GS, false>(G, p, sh.ws[wave], changed, env_r, task, ch);
    // An episode that runs out in this step and restarts from a known task (pre_ok): its starting grid row and occupancy
    // words start moving now as well -- the row by LDS-DMA into the LAST scratch slot (free unless the wavefront has as
    // many changed envs as slots), the occupancy words into a register -- so the reset at the end of the step is stores
    // only.  One env per wavefront (the first); any other env that resets takes reset_rows_wave.
    int staged_leader = -1;
    uint32_t staged_occ = aux_w;   // (any value: a dead register of the input burst)
    if (boost) {
        constexpr int R = req_chunk<GS>();
        const uint64_t pm = __ballot(pre_ok && writer);
        if (pm != 0 && __builtin_popcountll(chg_mask) < R) {
            staged_leader = __builtin_ctzll(pm);
            const int t_task = __builtin_amdgcn_readlane(task, staged_leader);
#if IGW_UNTRACKED_DMA
            const int8_t* row0 = p.task_start + (size_t)t_task * STRIDE;
            glds16u(row0, 16u * (uint32_t)G.lane, sh.ws[wave].hist[R - 1]);
            if (G.lane < CHUNKS - WAVE) glds16u(row0 + 16 * WAVE, 16u * (uint32_t)G.lane, sh.ws[wave].aux[R - 1]);
#else
            const char* row = reinterpret_cast<const char*>(p.task_start + (size_t)t_task * STRIDE) + 16 * G.lane;
            glds16(row, sh.ws[wave].hist[R - 1]);
            if (G.lane < CHUNKS - WAVE) glds16(row + 16 * WAVE, sh.ws[wave].aux[R - 1]);
#endif
            if (G.lane < OCC_WORDS) staged_occ = gload(p.task_start_occ + (size_t)t_task * OCC_WORDS + G.lane);
        }
    }
    const TailParams tp = tail_params(kernarg_again<STEP_KERNARG_HEAD>(p));
#if IGW_EARLY_COUNTERS
    // The event counters of a wavefront with changed envs are added HERE and behind the physics, not at the end of the
    // step: a wavefront is not retired before its atomics are acknowledged, and at the end of the step that wait -- an L2
    // round trip on a line sixteen blocks share -- was the last thing two wavefronts in three did (same-box A/B with the
    // counters compiled out: -5 % walking, -3.7 % CDM, -3.2 % flying; profiles/r06_ab_variants.txt).  Issued early, the
    // acknowledgements arrive while the wavefront computes.  IGW_STAT_RESCANS: env-steps that changed a cell.
    if (chg_mask != 0 && tp.stats != nullptr && G.lane == 0)
        counter_add(tp.stats + (blockIdx.x & (IGW_STAT_STRIPES - 1)) * 8 + IGW_STAT_RESCANS, (unsigned long long)__builtin_popcountll(chg_mask));
#endif
    prio_at<true, 3>(boost);
    if (MODE == MODE_FLY) world_update<GS, MODE_FLY, true>(G, p, e, occ_s, mv, boost);
    else world_update<GS, MODE_WALK, true>(G, p, e, occ_s, mv, boost);
    finish_break(e, ch);
    stamp(p, 3);
    prio_at<true, 5>(boost);
    // Everything fetched early (break colour, start byte, the DMA of the changed envs) has to be in by now; the
    // physics had the time of its sub-steps to cover it.
    if (chg_mask != 0 || staged_leader >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and counts as consumed HERE: the counter is in order over loads and stores, so a first use at the end of
    // the step would wait for every store issued from now on (observations, histogram pieces) as well
    asm volatile("" : "+v"(start_val), "+v"(env_max_int), "+v"(staged_occ));
    // (the prefetched reset metadata too: raw words, decoded in tail_step)
    asm volatile("" : "+v"(pre.a), "+v"(pre.b), "+v"(pre.c), "+v"(pre.d));
    // The agent record is final (pose, inventory, step_no): its store is issued here -- behind that wait, so it does
    // not wait for it -- and completes in the shadow of the histogram update's LDS round trips, not at the very end of
    // the wave.  (A reset at the end of this step overwrites it: same lanes, same addresses, program order.)  The
    // output record goes out whole at the end of the step, with reward and done: written in two parts (observations
    // here, results there) the 64-byte records reached memory as partial lines and the launch took 2 % longer.
    if constexpr (GS >= 4) {
        if (active && G.gl < 4)   // the whole agent record in ONE store instruction: lane q of the quad stores piece q
            st4(reinterpret_cast<uint4*>(p.agent + env) + G.gl, agent_piece(G.gl, e.x, e.y, e.z, e.yaw, e.pitch, e.vy, env_pack_piece3(e)));
    } else if (writer) {
        env_store(e, p.agent + env);
    }
    stamp(p, 4);
    const int size_new = e.prev_size + syn_size_delta(ch, start_val);  // synthetic grid = grid - start (env.py:290)
    const bool need = size_new != e.prev_size;  // wrong_placement != 0 -> recompute (tasks/task.py:112)
#if IGW_EARLY_COUNTERS
    if (chg_mask != 0) {   // IGW_STAT_CHANGED: env-steps whose block count changed (see above: in front of the histogram update)
        const uint64_t m_need = __ballot(need && active && G.gl == 0);
        if (m_need != 0 && tp.stats != nullptr && G.lane == 0)
            counter_add(tp.stats + (blockIdx.x & (IGW_STAT_STRIPES - 1)) * 8 + IGW_STAT_CHANGED, (unsigned long long)__builtin_popcountll(m_need));
    }
#endif
    const int hmax = resolve_changes<GS, false>(G, p, sh.ws[wave], chg_mask, env_r, task, ch, start_val, occ_wave_s);
    int mi = e.max_int;
    if (changed) {
        if (need) {  // max_int = maximal_intersection(grid)
            mi = hmax;
            e.dirty = 0;
        } else {
            e.dirty = 1;  // the reference keeps its cached max_int here although the grid changed
        }
    }
    stamp(p, 5);
    prio_at<true, 6>(boost);
    // (KParams sits behind the preloaded head arguments: kernarg_again reads it at that offset of the kernarg segment)
    tail_step<GS, MODE, EXTRA>(G, kernarg_again<STEP_KERNARG_HEAD>(p), a, sh, wave, env, active, e, ch, task, env_max_int, size_new, mi, need, changed, grid_g, occ_s, boost, diag_m, pre, pre_ok, tp, size_in, staged_leader, staged_occ,
                              EXACT ? BlockShared<GS>::EPB : min(BlockShared<GS>::EPB, n_envs - (int)blockIdx.x * BlockShared<GS>::EPB));
}

// T fused walking steps, state resident in registers + LDS.  Actions: counter RNG (auto-reset on done), or -- with
// `actions` [T][N] -- the caller's, with optional per-step rewards / dones [T][N] out and the context's autoreset
// setting (igw_rollout_walking_actions).
struct RolloutIO {
    const int32_t* actions;
    float* rewards;
    uint8_t* dones;
    // flying (igw_rollout_flying_actions): [T][N][3], [T][N][2], [T][N], [T][N]
    const float* movement;
    const float* camera;
    const int32_t* inventory;
    const int32_t* placement;
};
// __launch_bounds__(BLOCK, 3): left alone the loop keeps 196 registers live (2 waves per SIMD); three waves per SIMD
// (168 registers) measured fastest: 6.1 G against 5.4 G; four (128, 39 spilled) 5.9 G.  Groups of 16+ lanes serve
// batches of at most 4,096 envs (one wavefront per SIMD): no bound worth a spill there.
template <int GS, int MODE = MODE_WALK>
__global__ __launch_bounds__(BLOCK, ((GS == 4 || GS == 8) ? 3 : 2)) void rollout_kernel(KParams p, long long T, unsigned long long seed,
                                                        long long t0, long long env_offset, RolloutIO io) {
    __shared__ BlockShared<GS> sh;
    const Grp<GS> G;
    TrigCtx trig;
    trig.lut = trig_lut();
    const int wave = threadIdx.x / WAVE;
    const int slot = threadIdx.x / GS;
    const int env = blockIdx.x * BlockShared<GS>::EPB + slot;
    const int wave_env0 = blockIdx.x * BlockShared<GS>::EPB + wave * BlockShared<GS>::EPW;
    const bool active = env < p.n_envs;
    uint32_t* occ_wave_s = sh.occ + wave * BlockShared<GS>::EPW * OCC_PITCH;
    uint32_t* occ_s = sh.occ + slot * OCC_PITCH;
    if (wave_env0 >= p.n_envs) return;
    {
        OccStage<GS> st;
        occ_issue<GS>(p.occ, G, active ? env : p.n_envs - 1, st);
        occ_commit<GS>(G, st, occ_s, occ_wave_s);
    }
    Env e = {};
    int task = 0, env_max_int = 0;
    bool has_start = false;
    const TaskMeta* meta = nullptr;
    int8_t* grid_g = p.grid + (size_t)(active ? env : 0) * STRIDE;
    if (active) {
        task = env_load(e, p.agent + env, p.aux + env);
        meta = p.task_meta + task;
        has_start = meta->has_start != 0;
        env_max_int = meta->env_max_int;
    }
    wave_sync();
    unsigned long long n_changed = 0, n_resets = 0, n_updates = 0;
    StepOut o;
    o.reward = 0.0; o.done = false;
    bool last_reset = false;
    for (long long t = 0; t < T; t++) {
        // the parameters are read again from the kernarg segment in every phase of every step (see kernarg_again):
        // held in scalar registers over the whole loop they did not fit (121 of them parked in vector lanes)
        const KParams& p1 = kernarg_again(p);
        CellChange ch;
        ch.idx = -1; ch.bit = 0; ch.old_val = ch.new_val = 0; ch.occ_word = 0;
        Motion mv = {0.0, 0.0, 0.0};
        int size_new = 0, start_val = 0;
        bool need = false;
        if (active) {
            e.step_no = min(e.step_no + 1, 65535);
            if constexpr (MODE == MODE_FLY) {  // parse_flying_action, core/world.py:416-432 (as in step_kernel)
                const size_t at = (size_t)t * p1.n_envs + env;
                const int placement = io.placement[at];
                int inventory = io.inventory[at];
                double f[5] = {(double)io.movement[3 * at], (double)io.movement[3 * at + 1], (double)io.movement[3 * at + 2],
                               (double)io.camera[2 * at], (double)io.camera[2 * at + 1]};
                bool bad = false;
                if ((unsigned)inventory > 6u) { inventory = 0; bad = true; }
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    if (i < 3 ? !__builtin_isfinite(f[i]) : !camera_ok(f[i])) { f[i] = 0.0; bad = true; }
                }
                if (bad && G.gl == 0) stat_add(p1.stats, IGW_STAT_BAD_ACTION, 1);
                ch = world_act<GS, MODE_FLY>(G, p1, e, occ_s, grid_g, trig, f[0], f[1], f[2], inventory, f[3], f[4],
                                             placement == 2, placement == 1, mv, false, sh.ws[wave].hist[0]);
            } else {
                const int action = io.actions ? io.actions[(size_t)t * p1.n_envs + env]
                                              : rng_action18(seed, (uint64_t)(env_offset + env), (uint64_t)(t0 + t));
                const WalkAct w = parse_walking_discrete(action);
                ch = world_act<GS, MODE_WALK>(G, p1, e, occ_s, grid_g, trig, w.s0, w.s1, w.dy, w.inventory, w.cam0, w.cam1,
                                              w.remove, w.add, mv, false, sh.ws[wave].hist[0]);
            }
            if (ch.idx >= 0 && has_start) start_val = p1.task_start[(size_t)task * STRIDE + ch.idx];
        }
        const bool changed = active && ch.idx >= 0;
        const KParams& p2 = kernarg_again(p);
        const uint64_t chg_mask = prefetch_changes<GS, true>(G, p2, sh.ws[wave], changed, env, task, ch);
        if (active) {
            world_update<GS, MODE>(G, p2, e, occ_s, mv);
            finish_break(e, ch);
        }
        const int hmax = resolve_changes<GS, true>(G, p2, sh.ws[wave], chg_mask, env, task, ch, start_val);
        if (active) {
            size_new = e.prev_size + syn_size_delta(ch, start_val);
            need = size_new != e.prev_size;
        }
        int mi = e.max_int;
        if (changed) {
            if (need) {
                mi = hmax;
                e.dirty = 0;
            } else {
                e.dirty = 1;
            }
        }
        const KParams& p3 = kernarg_again(p);
        bool do_reset = false;
        if (active) {
            const TailParams tp3 = {p3.max_steps, p3.size_reward, p3.autoreset, p3.right_scale, p3.wrong_scale, nullptr, nullptr, nullptr, nullptr, nullptr};
            o = finish_step(tp3, e, env_max_int, size_new, mi);
            n_changed += need;
            n_updates += changed;
            do_reset = o.done && ((MODE == MODE_WALK && io.actions == nullptr) || p3.autoreset);
            last_reset = do_reset;
            if (G.gl == 0) {
                if (io.rewards) io.rewards[(size_t)t * p3.n_envs + env] = (float)o.reward;
                if (io.dones) io.dones[(size_t)t * p3.n_envs + env] = o.done ? 1 : 0;
            }
            // colours go to HBM right away (a later break of this launch reads them)
            if (ch.idx >= 0 && !do_reset && G.gl == 0) grid_g[ch.idx] = (int8_t)ch.new_val;
        }
        uint32_t ep = 0;
        int generated_size = -1;
        if (active && do_reset) {
            ep = next_task(p3, env, e, task);
            meta = p3.task_meta + task;
            has_start = !p3.rt_enabled && meta->has_start != 0;
            env_max_int = p3.rt_enabled ? 0 : meta->env_max_int;
        }
        wave_sync();
        resolve_resets<GS, true>(G, p3, do_reset, env, task, has_start, ep, occ_wave_s,
                                 reinterpret_cast<int8_t*>(&sh.ws[wave]), generated_size);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        if (active && do_reset) {
            reset_env_regs(e, meta, false, generated_size);
            n_resets++;
        }
    }
    if (!active) return;
    wave_sync();
    for (int w = G.gl; w < OCC_WORDS; w += GS) p.occ[(size_t)env * OCC_WORDS + w] = occ_s[OCC_VAR0 + w];
    if (G.gl == 0) {
        out_store(p.out + env, e, last_reset, (float)o.reward, o.done);
        env_store(e, p.agent + env);
        aux_store(e, task, p.aux + env);
        if (n_changed) stat_add(p.stats, IGW_STAT_CHANGED, n_changed);
        if (n_resets) stat_add(p.stats, IGW_STAT_RESETS, n_resets);
        if (n_updates) stat_add(p.stats, IGW_STAT_RESCANS, n_updates);
        stat_add(p.stats, IGW_STAT_STEPS, (unsigned long long)T);
    }
}

// GridWorld.reset for the masked envs: one wavefront per env (rows move coalesced)
__global__ __launch_bounds__(BLOCK) void reset_kernel(KParams p, const uint8_t* mask, int keep_size) {
    __shared__ alignas(16) int8_t row_s[WAVES_PER_BLOCK][STRIDE + IGW_LEVEL_INDEX_BYTES];  // RandomTasks generator scratch
    const int env = blockIdx.x * WAVES_PER_BLOCK + threadIdx.x / WAVE;
    if (env >= p.n_envs) return;
    if (mask && !mask[env]) return;
    Env e;
    int task = env_load(e, p.agent + env, p.aux + env);
    const uint32_t ep = next_task(p, env, e, task);
    const TaskMeta* meta = p.task_meta + task;
    int generated_size = -1;
    bool has_start = false;
    if (p.rt_enabled) generated_size = sample_random_task_wave(p, env, ep, row_s[threadIdx.x / WAVE]);
    else has_start = meta->has_start != 0;
    reset_rows_wave(p, env, task, has_start, nullptr);
    reset_env_regs(e, meta, keep_size != 0, generated_size);
    if (__lane_id() == 0) {
        out_store(p.out + env, e, true, 0.f, false);
        env_store(e, p.agent + env);
        aux_store(e, task, p.aux + env);
        if (p.traj && env < p.traj_n)  // the new episode's slot of the log starts empty
            *reinterpret_cast<int4*>(p.traj_heads + ((size_t)env * 2 + ((ep + 1) & 1)) * 4) = make_int4(task, 0, (int)(ep + 1), 0);
    }
}

__global__ void fill_actions_kernel(int32_t* actions, long long n_envs, long long n_steps, long long t0,
                                    unsigned long long seed, long long env_offset) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_envs * n_steps) return;
    const long long t = i / n_envs, env = i % n_envs;
    actions[i] = rng_action18(seed, (uint64_t)(env_offset + env), (uint64_t)(t0 + t));
}

// ---------------------------------------------------------------- Task.__init__ on device

struct PrepShared {
    alignas(16) int8_t a[WAVES_PER_BLOCK][STRIDE];  // target / synthetic target
    alignas(16) int8_t b[WAVES_PER_BLOCK][STRIDE];  // starting grid / evaluation grid
    alignas(16) int8_t c[WAVES_PER_BLOCK][STRIDE];  // full grid
    uint32_t hist[WAVES_PER_BLOCK][HIST_PAD];
};

__device__ inline void zero_row_lds_wave(int8_t* dst_s) {
    uint4* d = reinterpret_cast<uint4*>(dst_s);
    for (int c = __lane_id(); c < CHUNKS; c += WAVE) d[c] = make_uint4(0, 0, 0, 0);
}

// one wavefront per task row
__global__ __launch_bounds__(BLOCK) void prepare_tasks_kernel(KParams p, int first, int n,
                                                              const int8_t* user_target, const int8_t* start,
                                                              const int8_t* full_grid, const uint8_t* invariant,
                                                              const double* init_pose) {
    __shared__ PrepShared sh;
    const int wave = threadIdx.x / WAVE, lane = __lane_id();
    const int i = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (i >= n) return;
    const int task = first + i;
    int8_t* T = sh.a[wave];
    int8_t* S = sh.b[wave];
    int8_t* F = sh.c[wave];
    row_to_lds_wave(T, user_target + (size_t)i * STRIDE);
    if (start) row_to_lds_wave(S, start + (size_t)i * STRIDE);
    else zero_row_lds_wave(S);
    if (full_grid) row_to_lds_wave(F, full_grid + (size_t)i * STRIDE);
    wave_sync();
    {   // block ids are 0..7 (env.py:85: Box(low=-1, high=7); 0..6 in a starting grid): a cell with another id is read as empty, so that target
        // size, bounding boxes and colour index of the row stay consistent, and the row is counted
        bool odd = false;
        for (int c = lane; c < CELLS; c += WAVE) {
            if ((unsigned)T[c] > 7u) { T[c] = 0; odd = true; }
            if ((unsigned)S[c] > 6u) { S[c] = 0; odd = true; }   // (a starting block is taken off the inventory: ids 1..6, env.py:243-246)
            if (full_grid && (unsigned)F[c] > 7u) { F[c] = 0; odd = true; }
        }
        if (__ballot(odd) && lane == 0) stat_add(p.stats, IGW_STAT_BAD_TASK, 1);
        wave_sync();
    }
    // GridWorld.max_int at reset = user task on the starting grid (env.py:241); the user task's
    // admissible set comes from full_grid when given (task.py:63-66), or is [(0,0)] if not invariant
    const bool inv = invariant ? invariant[i] != 0 : true;
    int nnz_u = 0, nnz_s = 0;
    const BBox bu = row_bbox(full_grid ? F : T, nnz_u);
    int bb_user[4];
    if (inv) rot_bboxes(bu, nnz_u == 0, bb_user);
    else bb_user[0] = bb_user[1] = bb_user[2] = bb_user[3] = pack_bbox(0, 10, 0, 10);  // only (0, 0)
    const MiResult r = max_intersection_wave<false>(S, nullptr, T, sh.hist[wave], bb_user, inv ? 4 : 1);
    // starting grid -> table, inventory at reset (env.py:243-246)
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    for (int c = lane; c < CELLS; c += WAVE) {
        const int v = S[c];
        nnz_s += v != 0;
#pragma unroll
        for (int k = 0; k < 6; k++) cnt[k] += v == k + 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nnz_s += __shfl_xor(nnz_s, o, 64);
#pragma unroll
        for (int k = 0; k < 6; k++) cnt[k] += __shfl_xor(cnt[k], o, 64);
    }
    {
        const uint4* s = reinterpret_cast<const uint4*>(S);
        uint4* d = reinterpret_cast<uint4*>(const_cast<int8_t*>(p.task_start) + (size_t)task * STRIDE);
        for (int c = lane; c < CHUNKS; c += WAVE) d[c] = s[c];
        // occupancy bitmap of the starting grid in the padded 9 x 13 x 13 layout (igw_device.h): lane w packs
        // bits 32w .. 32w+31
        if (lane < OCC_WORDS) {
            uint32_t bits = 0;
            for (int k = 0; k < 32; k++) {
                const int i = lane * 32 + k, yv = i / OCC_LAYER, r = i % OCC_LAYER, xp = r / 13, zp = r % 13;
                if (yv < IGW_GRID_Y && xp >= 1 && xp <= 11 && zp >= 1 && zp <= 11 &&
                    S[yv * LEVEL + (xp - 1) * 11 + (zp - 1)] != 0)
                    bits |= 1u << k;
            }
            const_cast<uint32_t*>(p.task_start_occ)[(size_t)task * OCC_WORDS + lane] = bits;
        }
    }
    // synthetic target = target - starting grid (env.py:227-231), always invariant, no full grid
    wave_sync();
    for (int c = lane; c < STRIDE; c += WAVE) T[c] = (int8_t)(c < CELLS ? T[c] - S[c] : 0);
    wave_sync();
    int nnz_t = 0;
    const BBox bs = row_bbox(T, nnz_t);
    int bb_syn[4];
    rot_bboxes(bs, nnz_t == 0, bb_syn);
    {
        const uint4* s = reinterpret_cast<const uint4*>(T);
        uint4* d = reinterpret_cast<uint4*>(const_cast<int8_t*>(p.task_target) + (size_t)task * STRIDE);
        for (int c = lane; c < CHUNKS; c += WAVE) d[c] = s[c];
    }
    // the colour index of the synthetic target: what the step kernels vote from (sh.hist[wave] is free again)
    build_level_index_wave(T, bb_syn, reinterpret_cast<uint8_t*>(sh.hist[wave]),
                           const_cast<uint8_t*>(p.task_index) + (size_t)task * IGW_TASK_INDEX_BYTES);
    if (lane == 0) {
        TaskMeta m;
        memset(&m, 0, sizeof(m));
        // initial pose (GridWorld.initialize_world, env.py:177-193).  The ray march and the physics index the world
        // around the agent without range checks (igw_device.h), which holds for |x|, |z| <= 10: anything else, or a
        // non-finite value, is replaced by the default pose and counted (IGW_STAT_BAD_POSE)
        bool pose_ok = true;
        for (int k = 0; k < 5; k++) {
            const double v = init_pose ? init_pose[5 * (size_t)i + k] : 0.0;
            const double lim = (k == 0 || k == 2) ? 10.0 : k == 1 ? 64.0 : 1e6;
            pose_ok = pose_ok && __builtin_isfinite(v) && __builtin_fabs(v) <= lim;
            m.pose[k] = v;
        }
        if (!pose_ok) {
            for (int k = 0; k < 5; k++) m.pose[k] = 0.0;
            stat_add(p.stats, IGW_STAT_BAD_POSE, 1);
        }
        m.target_size = (int16_t)nnz_t;
        m.env_max_int = (int16_t)r.max_int;
        for (int k = 0; k < 4; k++) {
            m.bbox[4 * k + 0] = (int8_t)(bb_syn[k] & 0xff);
            m.bbox[4 * k + 1] = (int8_t)((bb_syn[k] >> 8) & 0xff);
            m.bbox[4 * k + 2] = (int8_t)((bb_syn[k] >> 16) & 0xff);
            m.bbox[4 * k + 3] = (int8_t)((bb_syn[k] >> 24) & 0xff);
        }
        for (int k = 0; k < 6; k++) m.inv_init[k] = (int16_t)(20 - cnt[k]);  // env.py:243-246: unbounded in the reference, >= -1069 here
        m.has_start = nnz_s != 0;
        const_cast<TaskMeta*>(p.task_meta)[task] = m;
    }
}

// stateless Task(target, full_grid, invariant) evaluated on grid (tasks/task.py:121-161)
__global__ __launch_bounds__(BLOCK) void task_eval_kernel(int n, const int8_t* target, const int8_t* grid,
                                                          const int8_t* full_grid, const uint8_t* invariant,
                                                          int32_t* max_int, int32_t* argmax,
                                                          int32_t* target_size) {
    __shared__ PrepShared sh;
    const int wave = threadIdx.x / WAVE, lane = __lane_id();
    const int i = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (i >= n) return;
    int8_t* T = sh.a[wave];
    int8_t* Gd = sh.b[wave];
    int8_t* F = sh.c[wave];
    row_to_lds_wave(T, target + (size_t)i * STRIDE);
    row_to_lds_wave(Gd, grid + (size_t)i * STRIDE);
    if (full_grid) row_to_lds_wave(F, full_grid + (size_t)i * STRIDE);
    wave_sync();
    const bool inv = invariant ? invariant[i] != 0 : true;
    int nnz_t = 0, nnz_f = 0;
    const BBox bt = row_bbox(T, nnz_t);
    BBox bu = bt;
    int nnz_u = nnz_t;
    if (full_grid) { bu = row_bbox(F, nnz_f); nnz_u = nnz_f; }
    int bb[4];
    if (inv) rot_bboxes(bu, nnz_u == 0, bb);
    else bb[0] = bb[1] = bb[2] = bb[3] = pack_bbox(0, 10, 0, 10);
    const MiResult r = max_intersection_wave<true>(Gd, nullptr, T, sh.hist[wave], bb, inv ? 4 : 1);
    if (lane == 0) {
        if (max_int) max_int[i] = r.max_int;
        if (argmax) { argmax[3 * i] = r.arg_dx; argmax[3 * i + 1] = r.arg_dz; argmax[3 * i + 2] = r.arg_rot; }
        if (target_size) target_size[i] = nnz_t;
    }
}

}  // namespace igw

// ================================================================= C ABI

using namespace igw;

struct igw_ctx {
    igw_config cfg;
    KParams kp;
    bool bound;
    int gs;
};

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) return fail(IGW_ERR_HIP, #expr ": %s", hipGetErrorString(_e)); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

// The walking trig table goes to constant memory once per device and process, at igw_create (the step
// entry points never copy or synchronise, so they stay legal inside a HIP-graph capture).
static std::mutex g_lut_mutex;
static bool g_lut_done[64];

static hipError_t upload_lut(int dev) {
    std::lock_guard<std::mutex> lock(g_lut_mutex);
    if (dev >= 0 && dev < 64 && g_lut_done[dev]) return hipSuccess;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(IGW_TRIG_LUT_DEV), IGW_TRIG_LUT_HOST, sizeof(IGW_TRIG_LUT_HOST));
    if (e == hipSuccess && dev >= 0 && dev < 64) g_lut_done[dev] = true;
    return e;
}

// Self-check of the general trig (igw_trig.h) ON THE DEVICE: the public functions' results, and whether a quick
// evaluation accepted a value that differs from the accurate one (it never may).  tests/test_gpu_flying.py compares
// the results with the host compile of the same header (correctly rounded on both sides).
__global__ void debug_trig_kernel(int64_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ sin_out,
                                  double* __restrict__ cos_out, double* __restrict__ atan_out, uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b[i];
    double s = 0.0, c = 0.0, qs = 0.0, qc = 0.0, as = 0.0, ac = 0.0, qa;
    const bool in_range = __builtin_fabs(x) < 0x1p20;   // igw_sincos is specified for |x| < 2^20 (the camera bound keeps it there)
    bool ok_sc = false;
    if (in_range) {
        igw_sincos(x, &s, &c);
        ok_sc = x != 0.0 && igw_sincos_quick(x, &qs, &qc);
        igw_sincos_accurate(x, &as, &ac);
    }
    const double at = igw_atan2(x, y);
    const bool ok_at = x != 0.0 && y != 0.0 && igw_atan2_quick(x, y, &qa);
    const double aa = (x != 0.0 && y != 0.0) ? igw_atan2_accurate(x, y) : at;
    sin_out[i] = s; cos_out[i] = c; atan_out[i] = at;
    flags[i] = (ok_sc ? 1 : 0) | ((ok_sc && (qs != as || qc != ac)) ? 2 : 0) | (ok_at ? 4 : 0) |
               ((ok_at && (qa != aa || __builtin_signbit(qa) != __builtin_signbit(aa))) ? 8 : 0);
}

extern "C" {

int igw_version(void) { return IGW_VERSION; }
#ifndef IGW_BUILD_ID
#define IGW_BUILD_ID "igw-build-id:unstamped"
#endif
// (the string carries a marker so that build.py can read the id of a library file without loading it)
const char* igw_build_id(void) { return &IGW_BUILD_ID[sizeof("igw-build-id:") - 1]; }
const char* igw_last_error(void) { return g_err; }

int igw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int igw_create(const igw_config* cfg, igw_ctx** out) {
    if (!cfg || !out) return fail(IGW_ERR_INVALID, "igw_create: null argument");
    *out = nullptr;
    if (cfg->num_envs < 1 || cfg->num_tasks < 1) return fail(IGW_ERR_INVALID, "igw_create: num_envs / num_tasks must be >= 1");
    if (cfg->max_steps < 1 || cfg->max_steps > 65534) return fail(IGW_ERR_INVALID, "igw_create: max_steps must be in 1..65534");
    if (cfg->action_space != IGW_WALKING_DISCRETE && cfg->action_space != IGW_FLYING &&
        cfg->action_space != IGW_WALKING_DICT)
        return fail(IGW_ERR_INVALID, "igw_create: unknown action_space");
    int gs = cfg->lanes_per_env;
    if (gs == 0) {
        // auto: the width measured fastest for the batch size (include/igw.h).  Four lanes per env (coordinate-split
        // ray march, axis-split collide) win from 32,768 envs up to 1,048,576; smaller batches need wider groups to
        // put enough wavefronts on the chip.
        gs = cfg->num_envs <= IGW_AUTO_32_MAX ? 32 : cfg->num_envs <= IGW_AUTO_16_MAX ? 16
           : cfg->num_envs <= IGW_AUTO_8_MAX ? 8 : 4;
    }
#ifndef IGW_DIAG
    if (cfg->reserved != 0)
        return fail(IGW_ERR_INVALID, "igw_create: reserved must be 0 (ablation switches exist only in the IGW_DIAG build)");
#endif
    if (gs != 64 && gs != 32 && gs != 16 && gs != 8 && gs != 4 && gs != 2 && gs != 1)
        return fail(IGW_ERR_INVALID, "igw_create: lanes_per_env must be 0 or a power of two in 1..64");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return fail(IGW_ERR_NO_DEVICE, "igw_create: no HIP device available (the HIP path has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= n) return fail(IGW_ERR_INVALID, "igw_create: device ordinal out of range");
    igw_ctx* c = new (std::nothrow) igw_ctx();
    if (!c) return fail(IGW_ERR_INVALID, "igw_create: out of memory");
    c->cfg = *cfg;
    c->bound = false;
    c->gs = gs;
    memset(&c->kp, 0, sizeof(c->kp));
    c->kp.n_envs = cfg->num_envs;
    c->kp.select_and_place = cfg->select_and_place;
    c->kp.size_reward = cfg->size_reward;
    c->kp.max_steps = cfg->max_steps;
    c->kp.autoreset = cfg->autoreset;
    c->kp.debug = cfg->reserved;
    c->kp.env_base = cfg->env_index_base;
    c->kp.right_scale = cfg->right_placement_scale;
    c->kp.wrong_scale = cfg->wrong_placement_scale;
    {
        DeviceGuard g(cfg->device);
        if (!g.ok) { delete c; return fail(IGW_ERR_HIP, "igw_create: hipSetDevice failed"); }
        hipError_t e = upload_lut(cfg->device);
        if (e != hipSuccess) { delete c; return fail(IGW_ERR_HIP, "igw_create: LUT upload: %s", hipGetErrorString(e)); }
    }
    *out = c;
    return IGW_OK;
}

int igw_set_task_sampling(igw_ctx* ctx, int32_t enabled, uint64_t seed, int32_t n_tasks) {
    if (!ctx) return fail(IGW_ERR_INVALID, "igw_set_task_sampling: null context");
    if (enabled) {
        if (!ctx->bound) return fail(IGW_ERR_UNBOUND, "igw_set_task_sampling: buffers not bound");
        if (n_tasks > ctx->cfg.num_tasks) return fail(IGW_ERR_INVALID, "igw_set_task_sampling: n_tasks exceeds the task table");
        if (ctx->kp.rt_enabled) return fail(IGW_ERR_INVALID, "igw_set_task_sampling: the RandomTasks generator is enabled on this context");
    }
    ctx->kp.sample_tasks = enabled ? 1 : 0;
    ctx->kp.n_tasks = n_tasks > 0 ? n_tasks : ctx->cfg.num_tasks;
    ctx->kp.sample_seed = seed;
    return IGW_OK;
}

int igw_set_random_tasks(igw_ctx* ctx, int32_t enabled, uint64_t seed, int32_t max_blocks, int32_t height_levels,
                         int32_t max_dist, int32_t num_colors, void* stream) {
    if (!ctx) return fail(IGW_ERR_INVALID, "igw_set_random_tasks: null context");
    if (!enabled) {
        ctx->kp.rt_enabled = 0;
        return IGW_OK;
    }
    if (!ctx->bound) return fail(IGW_ERR_UNBOUND, "igw_set_random_tasks: buffers not bound");
    if (ctx->cfg.num_tasks < ctx->cfg.num_envs) return fail(IGW_ERR_INVALID, "igw_set_random_tasks: needs one task row per env (num_tasks >= num_envs)");
    if (ctx->kp.sample_tasks) return fail(IGW_ERR_INVALID, "igw_set_random_tasks: task sampling is enabled on this context");
    if (max_blocks < 1 || height_levels < 1 || height_levels > IGW_GRID_Y || max_dist < 1 || num_colors < 1 || num_colors > 6)
        return fail(IGW_ERR_INVALID, "igw_set_random_tasks: need max_blocks >= 1, 1 <= height_levels <= 9, max_dist >= 1, 1 <= num_colors <= 6");
    DeviceGuard guard(ctx->cfg.device);
    if (!guard.ok) return fail(IGW_ERR_HIP, "igw_set_random_tasks: hipSetDevice failed");
    // generated tasks have an empty starting grid: the start rows of the envs' own task rows are zeroed once
    const size_t n = (size_t)ctx->cfg.num_envs;
    HIP_TRY(hipMemsetAsync(const_cast<int8_t*>(ctx->kp.task_start), 0, n * STRIDE, (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(const_cast<uint32_t*>(ctx->kp.task_start_occ), 0, n * OCC_WORDS * sizeof(uint32_t), (hipStream_t)stream));
    ctx->kp.rt_enabled = 1;
    ctx->kp.rt_max_blocks = max_blocks;
    ctx->kp.rt_levels = height_levels;
    ctx->kp.rt_max_dist = max_dist > 10 ? 10 : max_dist;  // 10 already covers the plane from any first block
    ctx->kp.rt_colors = num_colors;
    ctx->kp.sample_seed = seed;
    return IGW_OK;
}

int igw_set_trajectory_log(igw_ctx* ctx, void* records, int32_t* heads, int32_t n_logged, int32_t capacity) {
    if (!ctx) return fail(IGW_ERR_INVALID, "igw_set_trajectory_log: null context");
    if (!records || !heads || n_logged <= 0) {
        ctx->kp.traj = nullptr;
        ctx->kp.traj_heads = nullptr;
        ctx->kp.traj_n = 0;
        return IGW_OK;
    }
    if (!ctx->bound) return fail(IGW_ERR_UNBOUND, "igw_set_trajectory_log: buffers not bound");
    if (n_logged > ctx->cfg.num_envs || capacity < 1) return fail(IGW_ERR_INVALID, "igw_set_trajectory_log: n_logged / capacity out of range");
    if (((uintptr_t)records | (uintptr_t)heads) & 15) return fail(IGW_ERR_INVALID, "igw_set_trajectory_log: buffers must be 16-byte aligned");
    ctx->kp.traj = reinterpret_cast<uint8_t*>(records);
    ctx->kp.traj_heads = heads;
    ctx->kp.traj_n = n_logged;
    ctx->kp.traj_cap = capacity;
    return IGW_OK;
}

int igw_debug_set_stamps(igw_ctx* ctx, uint64_t* stamps) {
    if (!ctx) return fail(IGW_ERR_INVALID, "igw_debug_set_stamps: null context");
#ifdef IGW_DIAG
    ctx->kp.stamps = reinterpret_cast<unsigned long long*>(stamps);
    return IGW_OK;
#else
    (void)stamps;
    return fail(IGW_ERR_INVALID, "igw_debug_set_stamps: only available in the IGW_DIAG build (libigw_hip_diag.so)");
#endif
}

int igw_destroy(igw_ctx* ctx) {
    delete ctx;
    return IGW_OK;
}

int igw_bind_buffers(igw_ctx* ctx, const igw_buffers* b) {
    if (!ctx || !b) return fail(IGW_ERR_INVALID, "igw_bind_buffers: null argument");
    if (!b->grid || !b->occ || !b->hist || !b->agent || !b->aux || !b->task_target || !b->task_start ||
        !b->task_start_occ || !b->task_meta || !b->out || !b->task_index)
        return fail(IGW_ERR_INVALID, "igw_bind_buffers: a required buffer is null");
    if (((uintptr_t)b->grid | (uintptr_t)b->occ | (uintptr_t)b->hist | (uintptr_t)b->agent | (uintptr_t)b->aux | (uintptr_t)b->out |
         (uintptr_t)b->task_target | (uintptr_t)b->task_start | (uintptr_t)b->task_start_occ | (uintptr_t)b->task_meta |
         (uintptr_t)b->task_index) & 15)
        return fail(IGW_ERR_INVALID, "igw_bind_buffers: grid / occ / hist / agent / aux / out / task buffers must be 16-byte aligned");
    KParams& k = ctx->kp;
    k.grid = b->grid;
    k.occ = b->occ;
    k.hist = b->hist;
    k.task_start_occ = b->task_start_occ;
    k.agent = reinterpret_cast<AgentRec*>(b->agent);
    k.aux = reinterpret_cast<AuxRec*>(b->aux);
    k.task_target = b->task_target;
    k.task_start = b->task_start;
    k.task_meta = reinterpret_cast<const TaskMeta*>(b->task_meta);
    k.out = reinterpret_cast<OutRec*>(b->out);
    k.stats = reinterpret_cast<unsigned long long*>(b->stats);
    k.task_index = b->task_index;
    ctx->bound = true;
    return IGW_OK;
}

#define CHECK_CTX(name)                                                        \
    if (!ctx) return fail(IGW_ERR_INVALID, name ": null context");             \
    if (!ctx->bound) return fail(IGW_ERR_UNBOUND, name ": buffers not bound"); \
    DeviceGuard guard(ctx->cfg.device);                                        \
    if (!guard.ok) return fail(IGW_ERR_HIP, name ": hipSetDevice failed")

#define DISPATCH_GS(gs, CALL)          \
    switch (gs) {                      \
        case 64: { constexpr int GS = 64; CALL; } break; \
        case 32: { constexpr int GS = 32; CALL; } break; \
        case 16: { constexpr int GS = 16; CALL; } break; \
        case 8: { constexpr int GS = 8; CALL; } break;   \
        case 4: { constexpr int GS = 4; CALL; } break;   \
        case 2: { constexpr int GS = 2; CALL; } break;   \
        default: { constexpr int GS = 1; CALL; } break;  \
    }

// the step kernel with or without the rarely used extras (RandomTasks generator, episode log); A0..A3 = the leading
// action-buffer arguments of the mode (see step_kernel)
#define LAUNCH_STEP(MODE, A0, A1, A2, A3)                                                                          \
    do {                                                                                                           \
        if (ctx->kp.rt_enabled || ctx->kp.traj) {                                                                  \
            DISPATCH_GS(ctx->gs, hipLaunchKernelGGL((step_kernel<GS, MODE, true>), dim3(env_blocks(ctx)), dim3(BLOCK), 0, \
                                                    (hipStream_t)stream, ctx->kp.occ, ctx->kp.agent, ctx->kp.aux,      \
                                                    (const void*)(A0), (const void*)(A1), (const void*)(A2), (const void*)(A3), ctx->kp, a)); \
        } else {                                                                                                   \
            DISPATCH_GS(ctx->gs, hipLaunchKernelGGL((step_kernel<GS, MODE, false>), dim3(env_blocks(ctx)), dim3(BLOCK), 0, \
                                                    (hipStream_t)stream, ctx->kp.occ, ctx->kp.agent, ctx->kp.aux,      \
                                                    (const void*)(A0), (const void*)(A1), (const void*)(A2), (const void*)(A3), ctx->kp, a)); \
        }                                                                                                          \
    } while (0)

static inline int env_blocks(const igw_ctx* ctx) {
    const int epb = BLOCK / ctx->gs;
    return (ctx->cfg.num_envs + epb - 1) / epb;
}

int igw_prepare_tasks(igw_ctx* ctx, int32_t first, int32_t n, const int8_t* user_target, const int8_t* start,
                      const int8_t* full_grid, const uint8_t* invariant, const double* init_pose, void* stream) {
    CHECK_CTX("igw_prepare_tasks");
    if (n < 0 || first < 0 || first + n > ctx->cfg.num_tasks) return fail(IGW_ERR_INVALID, "igw_prepare_tasks: task range outside the table");
    if (!user_target) return fail(IGW_ERR_INVALID, "igw_prepare_tasks: user_target is null");
    if (n == 0) return IGW_OK;
    const int blocks = (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    hipLaunchKernelGGL(prepare_tasks_kernel, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, ctx->kp, first, n,
                       user_target, start, full_grid, invariant, init_pose);
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_reset(igw_ctx* ctx, const uint8_t* mask, int32_t flags, void* stream) {
    CHECK_CTX("igw_reset");
    const int keep = (flags & IGW_RESET_KEEP_SIZE) ? 1 : 0;
    const int blocks = (ctx->cfg.num_envs + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    hipLaunchKernelGGL(reset_kernel, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, ctx->kp, mask, keep);
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_step_walking(igw_ctx* ctx, const int32_t* actions, void* stream) {
    CHECK_CTX("igw_step_walking");
    if (ctx->cfg.action_space != IGW_WALKING_DISCRETE) return fail(IGW_ERR_INVALID, "igw_step_walking: context was created for another action space");
    if (!actions) return fail(IGW_ERR_INVALID, "igw_step_walking: actions is null");
    ActIn a = {actions, nullptr, nullptr, nullptr, nullptr, nullptr};
    LAUNCH_STEP(MODE_WALK, actions, nullptr, nullptr, (uintptr_t)ctx->kp.n_envs);
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_step_flying(igw_ctx* ctx, const float* movement, const float* camera, const int32_t* inventory,
                    const int32_t* placement, void* stream) {
    CHECK_CTX("igw_step_flying");
    if (ctx->cfg.action_space != IGW_FLYING) return fail(IGW_ERR_INVALID, "igw_step_flying: context was created for another action space");
    if (!movement || !camera || !inventory || !placement) return fail(IGW_ERR_INVALID, "igw_step_flying: an action buffer is null");
    ActIn a = {nullptr, movement, camera, inventory, placement, nullptr};
    if (ctx->gs == 4 && !ctx->kp.rt_enabled && !ctx->kp.traj && ctx->kp.n_envs % (BLOCK / 4) == 0) {  // whole blocks: the EXACT variant
        hipLaunchKernelGGL((step_kernel<4, MODE_FLY, false, true>), dim3(env_blocks(ctx)), dim3(BLOCK), 0, (hipStream_t)stream, ctx->kp.occ,
                           ctx->kp.agent, ctx->kp.aux, (const void*)movement, (const void*)camera, (const void*)inventory,
                           (const void*)placement, ctx->kp, a);
    } else {
        LAUNCH_STEP(MODE_FLY, movement, camera, inventory, placement);
    }
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_step_walking_dict(igw_ctx* ctx, const uint8_t* buttons, const float* camera, void* stream) {
    CHECK_CTX("igw_step_walking_dict");
    if (ctx->cfg.action_space != IGW_WALKING_DICT) return fail(IGW_ERR_INVALID, "igw_step_walking_dict: context was created for another action space");
    if (!buttons || !camera) return fail(IGW_ERR_INVALID, "igw_step_walking_dict: an action buffer is null");
    if ((uintptr_t)buttons & 7) return fail(IGW_ERR_INVALID, "igw_step_walking_dict: buttons must be 8-byte aligned");
    ActIn a = {nullptr, nullptr, camera, nullptr, nullptr, buttons};
    LAUNCH_STEP(MODE_WALK_DICT, buttons, camera, nullptr, (uintptr_t)ctx->kp.n_envs);
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_rollout_walking(igw_ctx* ctx, int64_t T, uint64_t seed, int64_t t0, int64_t env_offset, void* stream) {
    CHECK_CTX("igw_rollout_walking");
    if (ctx->kp.traj) return fail(IGW_ERR_INVALID, "igw_rollout_walking: the episode log is enabled on this context (the fused loop does not write it)");
    if (ctx->cfg.action_space != IGW_WALKING_DISCRETE) return fail(IGW_ERR_INVALID, "igw_rollout_walking: context was created for another action space");
    if (T < 0) return fail(IGW_ERR_INVALID, "igw_rollout_walking: T < 0");
    if (T == 0) return IGW_OK;
    DISPATCH_GS(ctx->gs, hipLaunchKernelGGL(rollout_kernel<GS>, dim3(env_blocks(ctx)), dim3(BLOCK), 0,
                                            (hipStream_t)stream, ctx->kp, (long long)T, (unsigned long long)seed,
                                            (long long)t0, (long long)env_offset, RolloutIO{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}));
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_rollout_walking_actions(igw_ctx* ctx, const int32_t* actions, int64_t T, float* rewards, uint8_t* dones,
                                void* stream) {
    CHECK_CTX("igw_rollout_walking_actions");
    if (ctx->kp.traj) return fail(IGW_ERR_INVALID, "igw_rollout_walking_actions: the episode log is enabled on this context (the fused loop does not write it)");
    if (ctx->cfg.action_space != IGW_WALKING_DISCRETE) return fail(IGW_ERR_INVALID, "igw_rollout_walking_actions: context was created for another action space");
    if (T < 0 || (T > 0 && !actions)) return fail(IGW_ERR_INVALID, "igw_rollout_walking_actions: bad argument");
    if (T == 0) return IGW_OK;
    DISPATCH_GS(ctx->gs, hipLaunchKernelGGL(rollout_kernel<GS>, dim3(env_blocks(ctx)), dim3(BLOCK), 0,
                                            (hipStream_t)stream, ctx->kp, (long long)T, 0ull, 0ll, 0ll,
                                            RolloutIO{actions, rewards, dones, nullptr, nullptr, nullptr, nullptr}));
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_rollout_flying_actions(igw_ctx* ctx, const float* movement, const float* camera, const int32_t* inventory,
                               const int32_t* placement, int64_t T, float* rewards, uint8_t* dones, void* stream) {
    CHECK_CTX("igw_rollout_flying_actions");
    if (ctx->kp.traj) return fail(IGW_ERR_INVALID, "igw_rollout_flying_actions: the episode log is enabled on this context (the fused loop does not write it)");
    if (ctx->cfg.action_space != IGW_FLYING) return fail(IGW_ERR_INVALID, "igw_rollout_flying_actions: context was created for another action space");
    if (T < 0 || (T > 0 && (!movement || !camera || !inventory || !placement))) return fail(IGW_ERR_INVALID, "igw_rollout_flying_actions: bad argument");
    if (T == 0) return IGW_OK;
    DISPATCH_GS(ctx->gs, hipLaunchKernelGGL((rollout_kernel<GS, MODE_FLY>), dim3(env_blocks(ctx)), dim3(BLOCK), 0,
                                            (hipStream_t)stream, ctx->kp, (long long)T, 0ull, 0ll, 0ll,
                                            RolloutIO{nullptr, rewards, dones, movement, camera, inventory, placement}));
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_fill_actions_walking(igw_ctx* ctx, int32_t* actions, int64_t n_steps, int64_t t0, uint64_t seed,
                             int64_t env_offset, void* stream) {
    CHECK_CTX("igw_fill_actions_walking");
    if (!actions || n_steps < 0) return fail(IGW_ERR_INVALID, "igw_fill_actions_walking: bad argument");
    const long long total = (long long)ctx->cfg.num_envs * n_steps;
    if (total == 0) return IGW_OK;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(fill_actions_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, actions,
                       (long long)ctx->cfg.num_envs, (long long)n_steps, (long long)t0, (unsigned long long)seed,
                       (long long)env_offset);
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_debug_trig(int32_t device, int64_t n, const double* a, const double* b, double* sin_out, double* cos_out,
                   double* atan_out, uint8_t* flags, void* stream) {
    if (n < 0 || !a || !b || !sin_out || !cos_out || !atan_out || !flags) return fail(IGW_ERR_INVALID, "igw_debug_trig: bad argument");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1)
        return fail(IGW_ERR_NO_DEVICE, "igw_debug_trig: no HIP device available (the HIP path has no CPU fallback)");
    if (device < 0 || device >= cnt) return fail(IGW_ERR_INVALID, "igw_debug_trig: device ordinal out of range");
    if (n == 0) return IGW_OK;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(IGW_ERR_HIP, "igw_debug_trig: hipSetDevice failed");
    hipLaunchKernelGGL(debug_trig_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, n, a, b,
                       sin_out, cos_out, atan_out, flags);
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

int igw_task_eval(int32_t device, int32_t n, const int8_t* target, const int8_t* grid, const int8_t* full_grid,
                  const uint8_t* invariant, int32_t* max_int, int32_t* argmax, int32_t* target_size, void* stream) {
    if (n < 0 || !target || !grid) return fail(IGW_ERR_INVALID, "igw_task_eval: bad argument");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1)
        return fail(IGW_ERR_NO_DEVICE, "igw_task_eval: no HIP device available (the HIP path has no CPU fallback)");
    if (device < 0 || device >= cnt) return fail(IGW_ERR_INVALID, "igw_task_eval: device ordinal out of range");
    if (n == 0) return IGW_OK;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(IGW_ERR_HIP, "igw_task_eval: hipSetDevice failed");
    const int blocks = (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    hipLaunchKernelGGL(task_eval_kernel, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, n, target, grid,
                       full_grid, invariant, max_int, argmax, target_size);
    HIP_TRY(hipGetLastError());
    return IGW_OK;
}

}  // extern "C"
