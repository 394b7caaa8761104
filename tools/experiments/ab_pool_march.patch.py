"""Round-5 experiment (rejected: profiles/r05_ab_variants.txt): builds the pooled-ray-march variant of the step kernel.

    python tools/experiments/ab_pool_march.patch.py <igw_kernels.hip of commit 04358e1> <output copy>

Applies textual replacements to the ROUND-5 kernel source (it asserts on every anchor, so it stops on any other
revision) and writes the result to the OUTPUT path -- never in place: a patched csrc/ changes the library's build id
and marks every committed profile stale (tests/test_profiles.py)."""
import os
import sys

if len(sys.argv) != 3 or os.path.abspath(sys.argv[1]) == os.path.abspath(sys.argv[2]):
    raise SystemExit(__doc__)
p, dst = sys.argv[1], sys.argv[2]
s = open(p).read()


def rep(old, new, cnt=1):
    global s
    assert s.count(old) >= 1, old[:80]
    s = s.replace(old, new, cnt)


# 1. template parameter
rep("""template <int GS, int MODE, bool EXTRA, bool EXACT = false>
__global__ __launch_bounds__(BLOCK, (GS >= 4 ? 4 : 2)) void step_kernel(""",
    """template <int GS, int MODE, bool EXTRA, bool EXACT = false, bool POOL = false>
__global__ __launch_bounds__(BLOCK, (GS >= 4 ? 4 : 2)) void step_kernel(""")
rep("""    static_assert(!EXACT || MODE == MODE_FLY, "EXACT is the flying kernel's variant");
    const int n_envs = MODE != MODE_FLY ? (int)(uintptr_t)h_a3 : EXACT ? 0x7fffffff : p.n_envs;
    const bool active = EXACT || env < n_envs;""",
    """    static_assert(!EXACT || MODE == MODE_FLY, "EXACT is the flying kernel's variant");
    // POOL (whole blocks only: block barriers, so no wavefront may leave early): the block's ray marches are pooled into
    // as few of its wavefronts as hold them, see pooled_hit_test
    static_assert(!POOL || (GS == 4 && MODE == MODE_WALK && !EXTRA && BLOCK == 256), "POOL is the Discrete(18) four-lane kernel's variant");
    const int n_envs = MODE != MODE_FLY ? (int)(uintptr_t)h_a3 : EXACT ? 0x7fffffff : p.n_envs;
    const bool active = EXACT || POOL || env < n_envs;""")
rep("""    if (!EXACT && wave_env0 >= n_envs) return;
    if (IGW_DIAG_FLAG(p, 64)) return;  // diag 64: the empty launch (same grid, registers and LDS)
    stamp(p, 0);""",
    """    if (!EXACT && !POOL && wave_env0 >= n_envs) return;
    if (IGW_DIAG_FLAG(p, 64)) return;  // diag 64: the empty launch (same grid, registers and LDS)
    stamp(p, 0);""")
rep("""    if (ap.want_sight) h = hit_test<GS, true>(G, occ_s, e.x, e.y, e.z, ap.vx, ap.vy, ap.vz, boost, sh.ws[wave].hist[0]);
    ch = world_act_post<GS, false>(G, e, occ_s, grid_g, ap, h);""",
    """    if constexpr (POOL) h = pooled_hit_test(G, sh, wave, slot, ap.want_sight, e.x, e.y, e.z, ap.vx, ap.vy, ap.vz, boost);
    else if (ap.want_sight) h = hit_test<GS, true>(G, occ_s, e.x, e.y, e.z, ap.vx, ap.vy, ap.vz, boost, sh.ws[wave].hist[0]);
    ch = world_act_post<GS, false>(G, e, occ_s, grid_g, ap, h);""")

# 2. the pooled march, in front of the kernel
rep("""// __launch_bounds__(BLOCK, 4): four waves per SIMD, i.e. at most 128 VGPRs""",
    """// The block's ray marches pooled into as few wavefronts as hold them (four lanes per env, whole blocks).
// Under uniform Discrete(18) actions 8 of 18 actions look along the sight vector, so about 28 of a block's 64 envs
// march -- but the 40-sample loop costs a wavefront the same whether one of its sixteen envs marches or all do.  Here
// the marching envs of the BLOCK publish their rays in LDS (position + sight vector, 48 bytes), are numbered
// 0 .. M - 1, and pass k = rays 16 k .. 16 k + 15 is marched by ONE wavefront against the occupancy rows of the rays'
// envs (the rows of the whole block are in LDS anyway); usually two of the four wavefronts march and the other two
// leave their issue slots to them.  The marching wavefront of pass k is (k + block / 256) mod 4: the four blocks of
// a CU rotate, so every SIMD keeps its share.  Rays live in the last histogram slot of the scratch of wavefront k
// (idle until the changes are fetched), results in the aux slot of the env's own wavefront; three block barriers
// (LDS only: s_waitcnt lgkmcnt(0) + s_barrier, the global loads of the burst stay in flight).
__device__ inline void block_sync_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");
#endif
}
__device__ inline Hit pooled_hit_test(const Grp<4>& G, BlockShared<4>& sh, int wave, int slot, bool want, double x, double y,
                                      double z, double vx, double vy, double vz, bool boost) {
    constexpr int R = req_chunk<4>();
    static_assert(16 * 6 * 8 + 16 * 4 <= (HIST_ROW / 2) * 4 && 16 * 8 + 16 <= LVL_BYTES, "ray pool fits the scratch slots");
    Hit h;
    h.hit = false; h.have_prev = false;
    h.bx = h.by = h.bz = h.px = h.py = h.pz = 0;
    const uint64_t mm = __ballot(want && G.gl == 0);
    uint32_t* const cnt_s = sh.ws[0].aux[1];                       // [4] marching envs per wavefront
    if (G.lane == 0) cnt_s[wave] = (uint32_t)__builtin_popcountll(mm);
    block_sync_lds();
    const int c0 = (int)cnt_s[0], c1 = (int)cnt_s[1], c2 = (int)cnt_s[2], c3 = (int)cnt_s[3];
    const int total = c0 + c1 + c2 + c3;
    if (total == 0) return h;                                       // (block-uniform)
    const int base = (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
    if (want && G.gl == 0) {                                        // publish this env's ray as ray number s
        const int s = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
        double* ray = reinterpret_cast<double*>(sh.ws[s >> 4].hist[R - 1]) + 6 * (s & 15);
        ray[0] = x; ray[1] = y; ray[2] = z; ray[3] = vx; ray[4] = vy; ray[5] = vz;
        sh.ws[s >> 4].hist[R - 1][16 * 12 + (s & 15)] = (uint32_t)slot;
    }
    block_sync_lds();
    const int pass = (wave - (int)(blockIdx.x >> 8)) & 3;          // the pass this wavefront marches, if there is one
    if (16 * pass < total) {                                        // (wave-uniform)
        const int s = 16 * pass + (G.lane >> 2);
        const bool have = s < total;
        const int sl = have ? (s & 15) : 0;                         // idle quads march ray 0 of the pass again (discarded)
        const double* ray = reinterpret_cast<const double*>(sh.ws[pass].hist[R - 1]) + 6 * sl;
        const int eslot = (int)sh.ws[pass].hist[R - 1][16 * 12 + sl];
        const double rx = ray[0], ry = ray[1], rz = ray[2], rvx = ray[3], rvy = ray[4], rvz = ray[5];
        const Hit r = hit_test<4, true>(G, sh.occ + eslot * OCC_PITCH, rx, ry, rz, rvx, rvy, rvz, boost, sh.ws[wave].hist[0]);
        if (have && G.gl == 0) {                                    // to the env's own wavefront: two packed words
            const uint32_t bk = (uint32_t)(r.bx + 6) | ((uint32_t)(r.by + 4) << 8) | ((uint32_t)(r.bz + 6) << 16) |
                                ((uint32_t)r.hit << 24) | ((uint32_t)r.have_prev << 25);
            const uint32_t pk = (uint32_t)(r.px + 6) | ((uint32_t)(r.py + 4) << 8) | ((uint32_t)(r.pz + 6) << 16);
            *reinterpret_cast<uint2*>(&sh.ws[eslot >> 4].aux[0][2 * (eslot & 15)]) = make_uint2(bk, pk);
        }
    }
    block_sync_lds();
    if (want) {
        const uint2 w = *reinterpret_cast<const uint2*>(&sh.ws[wave].aux[0][2 * (slot & 15)]);
        h.hit = (w.x >> 24) & 1u; h.have_prev = (w.x >> 25) & 1u;
        h.bx = (int)(w.x & 0xff) - 6; h.by = (int)((w.x >> 8) & 0xff) - 4; h.bz = (int)((w.x >> 16) & 0xff) - 6;
        h.px = (int)(w.y & 0xff) - 6; h.py = (int)((w.y >> 8) & 0xff) - 4; h.pz = (int)((w.y >> 16) & 0xff) - 6;
    }
    wave_sync();   // (the aux slot is a DMA destination further down)
    return h;
}

// __launch_bounds__(BLOCK, 4): four waves per SIMD, i.e. at most 128 VGPRs""")

# 3. dispatch: pooled variant for Discrete(18) walking, four lanes, whole blocks, no extras
rep("""    ActIn a = {actions, nullptr, nullptr, nullptr, nullptr, nullptr};
    LAUNCH_STEP(MODE_WALK, actions, nullptr, nullptr, (uintptr_t)ctx->kp.n_envs);""",
    """    ActIn a = {actions, nullptr, nullptr, nullptr, nullptr, nullptr};
#ifndef IGW_NO_POOL
    if (ctx->gs == 4 && !ctx->kp.rt_enabled && !ctx->kp.traj && ctx->kp.n_envs % (BLOCK / 4) == 0 && !ctx->kp.debug) {  // whole blocks: the POOL variant
        hipLaunchKernelGGL((step_kernel<4, MODE_WALK, false, false, true>), dim3(env_blocks(ctx)), dim3(BLOCK), 0, (hipStream_t)stream,
                           ctx->kp.occ, ctx->kp.agent, ctx->kp.aux, (const void*)actions, (const void*)nullptr, (const void*)nullptr,
                           (const void*)(uintptr_t)ctx->kp.n_envs, ctx->kp, a);
    } else
#endif
    LAUNCH_STEP(MODE_WALK, actions, nullptr, nullptr, (uintptr_t)ctx->kp.n_envs);""")
open(dst, 'w').write(s)
