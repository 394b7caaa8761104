/*
 * igw.h -- C ABI of the MI355X-native IGLU gridworld step path (libigw_hip.so).
 *
 * Drop-in boundary for the reference's `GridWorld` env protocol on the
 * render=False / vector_state=True path (all `file:line` are relative to the
 * reference, iglu-contest/gridworld):
 *
 *   igw_prepare_tasks  <- GridWorld.set_task / set_task_generator + Task.__init__
 *                         (gridworld/env.py:155-175, gridworld/tasks/task.py:9-72)
 *                         + GridWorld.initialize_world pose (env.py:177-193)
 *   igw_reset          <- GridWorld.reset (env.py:206-261), SizeReward.reset (env.py:321-323)
 *   igw_step_walking   <- GridWorld.step (env.py:268-303) with World.step /
 *                         parse_walking_discrete_action (core/world.py:360-394, 434-456)
 *   igw_step_flying    <- same with parse_flying_action (core/world.py:416-432)
 *   igw_step_walking_dict <- same with parse_walking_action, discretize=False (core/world.py:396-414)
 *   igw_task_eval      <- Task.maximal_intersection / argmax_intersection
 *                         (tasks/task.py:121-161)
 *   igw_rollout_walking, igw_rollout_walking_actions, igw_rollout_flying_actions <- the loop of examples/run_env.py:18-26 fused on device
 *
 * Plain pointers and sizes only; every data pointer is a DEVICE pointer owned by
 * the caller (e.g. torch tensors) and must outlive the context.  All calls are
 * asynchronous on the given hipStream_t (pass it as void*; NULL = default stream),
 * never synchronise, never allocate, and return 0 on success or a negative
 * igw_status; igw_last_error() gives the message.  No exceptions cross the ABI.
 * One batch of N independent envs per context; the caller serialises calls on
 * one context.
 */
#ifndef IGW_H
#define IGW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGW_VERSION 5
/* igw_config.lanes_per_env == 0 picks the group width measured fastest on an MI355X for the batch size
 * (profiles/r02_sweep_lanes*.txt): 32 lanes per env up to IGW_AUTO_32_MAX envs, 16 up to IGW_AUTO_16_MAX, 8 up to
 * IGW_AUTO_8_MAX, 4 beyond (a launch then has between about 512 and 4,096 wavefronts for 1,024 SIMDs).  64 (one
 * wavefront per env), 2 and 1 lanes per env exist and are tested, but are never chosen automatically. */
#define IGW_AUTO_32_MAX 1024
#define IGW_AUTO_16_MAX 4096
#define IGW_AUTO_8_MAX 24576

/* dense voxel grid [y+1][x+5][z+5], int8 (env.py:34, 136-142) */
#define IGW_GRID_Y 9
#define IGW_GRID_X 11
#define IGW_GRID_Z 11
#define IGW_CELLS 1089
/* row stride of every [*, 9,11,11] int8 buffer: 1089 padded to a multiple of 16 B so a
 * wavefront moves one env's grid with 69 aligned dwordx4 accesses; pad bytes stay 0 */
#define IGW_GRID_STRIDE 1104
/* occupancy bitmap, 48 dwords = 192 B per env: the per-step working set of the physics, kept in sync with
 * `grid` by every kernel.  One bit per cell of the 9 x 13 x 13 box that pads each y level of the build zone
 * with one always-empty cell on every side in x and z: bit = (y+1)*169 + (x+6)*13 + (z+6) for the cell at
 * world (x, y, z), i.e. grid[y+1][x+5][z+5]; padding bits and bits >= 1521 stay 0 */
#define IGW_OCC_WORDS 48
/* persistent per-env vote histogram of maximal_intersection: for each of the 4 rotations the 11 x 11
 * admissible translations (bounding-box relative), uint16 counts, row padded to 512 entries */
#define IGW_HIST_ROW 512
/* colour index of a task's synthetic target: what the step kernels vote from when a cell changes.  Per y level one
 * IGW_LEVEL_INDEX_BYTES block:
 *    0 i8 bbox[4][4]   per rotation xmin, xmax, zmin, zmax of the synthetic target (copy of the metadata's)
 *   16 u8 offs[15]     start of each colour class in cells[]; class = colour + 7 for -7..-1, colour + 6 for 1..7
 *                      (synthetic colours = target - start with block ids 0..7); offs[14] = cells on this level;
 *                      byte 31 repeats offs[14] (an empty slice behind the last class)
 *   32 u8 cells[<=121] (x+5) << 4 | (z+5) of the level's target cells, sorted by colour class
 * written by igw_prepare_tasks and by the on-device RandomTasks generator */
#define IGW_LEVEL_INDEX_BYTES 160
#define IGW_TASK_INDEX_BYTES (IGW_GRID_Y * IGW_LEVEL_INDEX_BYTES)
/* bytes of per-env agent state (two records) , of the per-step output record and of per-task metadata (layouts below) */
#define IGW_AGENT_BYTES 64
#define IGW_AUX_BYTES 16
#define IGW_OUT_BYTES 64
#define IGW_TASK_META_BYTES 128
/* striped device counters: [IGW_STAT_STRIPES][8] uint64, stripe = 64 B */
#define IGW_STAT_STRIPES 64
#define IGW_STAT_CHANGED 0 /* env-steps whose block count changed (max_intersection recomputed) */
#define IGW_STAT_RESETS 1  /* auto-resets performed */
#define IGW_STAT_STEPS 2   /* env-steps executed: every step launch adds its num_envs (ONE atomic per launch: the first
                            * wavefront of block 0, stripe 0), the fused rollouts add T per env -- the device-side count of
                            * the work done */
#define IGW_STAT_RESCANS 3 /* histogram row updates (env-steps that changed a cell; each takes the row maximum) */
#define IGW_STAT_BAD_POSE 4   /* task rows whose init_pose was rejected (non-finite, |x| or |z| > 10, |y| > 64,
                               * |yaw| or |pitch| > 1e6) and replaced by the default pose */
#define IGW_STAT_BAD_ACTION 5 /* env-steps whose action was rejected and executed as a no-op component: non-finite
                               * movement / camera values, camera deltas beyond +-IGW_CAMERA_MAX, inventory /
                               * hotbar ids outside 0..6 (the reference raises ValueError there,
                               * core/world.py:354-355) */
#define IGW_STAT_BAD_TASK 6   /* task rows with a block id outside 0..7 in target or full grid (the ids of env.py:85)
                               * or outside 0..6 in the starting grid (a starting block is taken off the inventory of
                               * its colour, env.py:243-246; the reference raises IndexError for 7): such a cell is read
                               * as empty (0), so the row stays consistent (target size, boxes, colour index) */
/* largest |camera delta| per step (degrees; same bound as init_pose's yaw / pitch): the reference wraps yaw by
 * repeated subtraction of 360 (core/world.py:451-456), which never ends for a finite but huge value */
#define IGW_CAMERA_MAX 1e6

enum igw_status {
    IGW_OK = 0,
    IGW_ERR_INVALID = -1,   /* bad argument / config */
    IGW_ERR_NO_DEVICE = -2, /* no usable HIP device */
    IGW_ERR_HIP = -3,       /* a HIP call failed */
    IGW_ERR_UNBOUND = -4    /* buffers not bound */
};

enum igw_action_space {
    IGW_WALKING_DISCRETE = 0, /* Discrete(18), create_env default (env.py:333-338) */
    IGW_FLYING = 1,           /* Dict(movement, camera, inventory, placement) (env.py:71-78) */
    IGW_WALKING_DICT = 2      /* discretize=False: Dict of buttons + continuous camera (env.py:60-70) */
};

/* create_env kwargs that reach the step path (env.py:333-350) */
typedef struct igw_config {
    int32_t device;            /* HIP device ordinal */
    int32_t num_envs;          /* N */
    int32_t num_tasks;         /* capacity of the task table (>= 1) */
    int32_t action_space;      /* igw_action_space */
    int32_t select_and_place;  /* env.py:334 (default 1) */
    int32_t size_reward;       /* SizeReward wrapper, env.py:316-331 (default 1) */
    int32_t max_steps;         /* env.py:336 (default 250), 1..65534 */
    int32_t autoreset;         /* 0: caller resets on done (reference loop); 1: reset inside step */
    double right_placement_scale; /* env.py:335 */
    double wrong_placement_scale; /* env.py:337 */
    int32_t lanes_per_env;     /* 0 = automatic from num_envs (see IGW_AUTO_*_MAX); or 64, 32, ..., 1 */
    int32_t reserved;          /* must be 0 (ablation switches of the IGW_DIAG build) */
    int64_t env_index_base;    /* global index of env 0 of this context (rank offset, sub-batch offset): keys the
                                * on-device task samplers so shards and sub-batches draw different streams */
} igw_config;

/*
 * Agent state, IGW_AGENT_BYTES per env (Agent, core/world.py:8-29, + GridWorld.step_no): rewritten whole by every
 * step, one 16-byte piece per lane of the env's quad:
 *   0  f64 x, y, z        agent.position
 *   24 f64 yaw            agent.rotation[0]
 *   32 f64 pitch          agent.rotation[1]
 *   40 f64 vy             agent.dy            (leaks through reset)
 *   48 i16 inventory[6]   agent.inventory: 20 - (blocks of that colour in the world), i.e. -1069..20 -- the
 *                         reference's Python ints are unbounded (env.py:243-246: one decrement per starting block)
 *   60 u16 step_no        GridWorld.step_no   (saturates at 65535)
 *   62 u16 pack           bits 0-1 agent.time_int_steps code (0,1,2,3 = 2,4,8,12), bits 2-4 agent.active_block
 *                         (both leak through reset)
 * A fresh agent (Agent.__init__) is inventory 20 x 6, pack = 1 << 2 (time_int_steps 2, BLUE).
 *
 * Episode state, IGW_AUX_BYTES per env (the ints GridWorld / Task / SizeReward keep per episode + the env's task):
 * read by every step, written only by the steps that change it (a block placed or broken, SizeReward's first
 * step, a reset):
 *   0  i16 size           SizeReward.size
 *   2  u16 prev_size      _synthetic_task.prev_grid_size (bits 0-14); bit 15: histogram changed while the
 *                         reference kept its cached max_int (a change with wrong_placement == 0)
 *   4  i16 max_int        _synthetic_task.max_int
 *   6  i16 target_size    _synthetic_task.target_size (copied from the task table by reset, so a step needs no
 *                         task-table access unless the grid changed)
 *   8  i32 task           the env's row of the task table (GridWorld._task)
 *   12 u32 episode        episodes started (every reset adds 1): keys the on-device task samplers and the
 *                         trajectory log
 *
 * Per-step outputs, IGW_OUT_BYTES per env (env.py:281-303), written whole by every step:
 *   0  f32 agentPos[5]    x, y, z, pitch, yaw
 *   20 f32 inventory[6]
 *   44 f32 compass        yaw - 180
 *   48 f32 reward         (float) of the double reward
 *   52 u8  done
 *   53 ..  zero
 *
 * Task metadata, IGW_TASK_META_BYTES per task (written by igw_prepare_tasks):
 *   0  f64 init_pose[5]   x, y, z, yaw, pitch (GridWorld.initial_position/rotation)
 *   40 i16 target_size    _synthetic_task.target_size
 *   42 i16 env_max_int    GridWorld.max_int at reset (user task on the starting grid)
 *   44 u8  has_start      starting grid not empty
 *   48 i8  bbox[4][4]     per rotation xmin, xmax, zmin, zmax of the synthetic target
 *   64 i16 inv_init[6]    inventory at reset (20 - blocks of that colour in the start grid)
 *   76 ..  zero
 */
typedef struct igw_buffers {
    /* state */
    int8_t* grid;          /* [N][IGW_GRID_STRIDE]   world grid (colours) == obs 'grid' */
    uint32_t* occ;         /* [N][IGW_OCC_WORDS]     occupancy bitmap of grid */
    uint16_t* hist;        /* [N][IGW_HIST_ROW]      vote histogram of (grid - start) against the synthetic target */
    void* agent;           /* [N][IGW_AGENT_BYTES] */
    void* aux;             /* [N][IGW_AUX_BYTES] */
    /* task table */
    int8_t* task_target;   /* [T][IGW_GRID_STRIDE] synthetic target = target - start (env.py:230) */
    int8_t* task_start;    /* [T][IGW_GRID_STRIDE] dense starting grid (env.py:226) */
    uint32_t* task_start_occ; /* [T][IGW_OCC_WORDS] its occupancy bitmap */
    void* task_meta;       /* [T][IGW_TASK_META_BYTES] */
    uint8_t* task_index;   /* [T][IGW_TASK_INDEX_BYTES] colour index of task_target (task table, 16-byte aligned) */
    /* per-step outputs (env.py:281-303): agentPos, inventory, compass, reward, done of every env in one record */
    void* out;             /* [N][IGW_OUT_BYTES] */
    uint64_t* stats;       /* [IGW_STAT_STRIPES][8], caller zeroes; may be NULL */
} igw_buffers;

typedef struct igw_ctx igw_ctx;

int igw_version(void);
/* Identity of the kernel build: a hash of the sources the library was compiled from (csrc/ + this header + the
 * compiler flags; gridworld_amd/build.py: source_hash).  Profiles under profiles/ carry the id of the library they
 * were taken with; bench.py marks a profile of another build as stale. */
const char* igw_build_id(void);
const char* igw_last_error(void);
/* number of visible HIP devices (0 if none / runtime unusable); does not create a context */
int igw_device_count(void);

int igw_create(const igw_config* cfg, igw_ctx** out);
int igw_destroy(igw_ctx* ctx);
/* diagnostic: device uint64 [waves][8] receiving in-kernel s_memtime phase stamps of the step kernels
 * (drains the wave's memory queues at every stamp, so never set it in a timed run); NULL disables */
int igw_debug_set_stamps(igw_ctx* ctx, uint64_t* stamps);
int igw_bind_buffers(igw_ctx* ctx, const igw_buffers* bufs);

/* Fills task-table rows [first, first+n): user_target / start / full_grid are device
 * int8 [n][IGW_GRID_STRIDE] (start, full_grid may be NULL = empty / absent), invariant is
 * device uint8[n] or NULL (= 1), init_pose device double[n][5] or NULL (= zeros). */
int igw_prepare_tasks(igw_ctx* ctx, int32_t first, int32_t n, const int8_t* user_target,
                      const int8_t* start, const int8_t* full_grid, const uint8_t* invariant,
                      const double* init_pose, void* stream);

/* Task generator on the device (CustomTasks.reset, gridworld/tasks/task_set.py:53-56): when enabled every
 * reset -- igw_reset and the auto-reset inside the step kernels -- first draws the env's task (aux record) uniformly from rows
 * [0, n_tasks) of the task table (n_tasks <= 0: the whole table) with a counter RNG keyed by (seed,
 * env_index_base + env, the env's episode counter).  Same distribution as the reference's np.random.choice, not the
 * same stream; replayable from a HIP graph (the key lives in device memory: the aux record's episode counter). */
int igw_set_task_sampling(igw_ctx* ctx, int32_t enabled, uint64_t seed, int32_t n_tasks);

/* RandomTasks.sample_task on the device (gridworld/tasks/task_set.py:135-157): when enabled every reset first
 * writes a freshly sampled target into the env's OWN task row (its task := env; needs num_tasks >=
 * num_envs; starting grid empty, the row's init pose is kept): per height level one block uniform over the
 * 11 x 11 plane, then max_blocks - 1 further blocks on distinct cells within Chebyshev distance max_dist of
 * it (all of them when fewer are free -- where the reference's rejection loop never ends), colours uniform in
 * 1..num_colors.  Sequential rejection sampling without replacement = a uniformly random subset, which is what
 * the kernel draws in parallel (counter RNG keyed as above): same distribution, not the same stream.
 * Mutually exclusive with igw_set_task_sampling.  Asynchronous on `stream` (zeroes the starting-grid rows). */
int igw_set_random_tasks(igw_ctx* ctx, int32_t enabled, uint64_t seed, int32_t max_blocks, int32_t height_levels,
                         int32_t max_dist, int32_t num_colors, void* stream);

/* Episode log on the device (what the reference's Logged wrapper collects per step, gridworld/wrappers.py:
 * 89-121, minus video): for envs [0, n_logged) every step writes one IGW_TRAJ_BYTES record at
 *   records[env][episode & 1][step_no - 1]          (two episodes of `capacity` steps per env, so a finished
 * episode stays readable while the next one is written; steps beyond capacity are not recorded) and keeps
 *   heads[env][episode & 1] = { task row, steps recorded, episode number, finished }  (int32 x 4) current;
 * `finished` is the done flag of the last recorded step (1: the episode in this slot is complete).
 * Record layout (little endian):
 *    0 f32 agentPos[5]   20 f32 reward   24 f32 compass   28 i16 inventory[6]
 *   40 u16 change: 0xffff, or cell (bits 0-10) | colour (bits 11-13)     42 u8 done
 *   43 u8 action space (igw_action_space, bits 0-1) | flying: inventory (bits 2-4), placement (bits 5-6), as executed
 *   44 walking: i32 action | flying: f32 movement[3], f32 camera[2] | walking Dict: u8 buttons[8], f32 camera[2]
 * records / heads NULL disables. */
#define IGW_TRAJ_BYTES 64
int igw_set_trajectory_log(igw_ctx* ctx, void* records, int32_t* heads, int32_t n_logged, int32_t capacity);

#define IGW_RESET_KEEP_SIZE 1 /* GridWorld.reset only (what set_task calls): SizeReward.size survives */
/* mask: device uint8[N] or NULL (= all envs) */
int igw_reset(igw_ctx* ctx, const uint8_t* mask, int32_t flags, void* stream);

int igw_step_walking(igw_ctx* ctx, const int32_t* actions /* [N] in 0..17 */, void* stream);
int igw_step_flying(igw_ctx* ctx, const float* movement /* [N][3] */, const float* camera /* [N][2] */,
                    const int32_t* inventory /* [N] 0..6 */, const int32_t* placement /* [N] 0..2 */,
                    void* stream);

/* walking with discretize=False (parse_walking_action, core/world.py:396-414): buttons uint8 [N][8] =
 * forward, back, left, right, jump, attack, use, hotbar(0..6), 8-byte aligned; camera float [N][2] */
int igw_step_walking_dict(igw_ctx* ctx, const uint8_t* buttons, const float* camera, void* stream);

/* T fused walking steps per env, actions = uniform Discrete(18) from a counter RNG keyed by
 * (seed, env_offset + env, t) for t = t0 .. t0+T-1; auto-reset on done regardless of cfg.autoreset. */
int igw_rollout_walking(igw_ctx* ctx, int64_t T, uint64_t seed, int64_t t0, int64_t env_offset,
                        void* stream);
/* The same fused loop over caller-supplied actions: T steps per env with actions[t][env] (int32 [T][N], 0..17) --
 * exactly T calls of igw_step_walking (the context's autoreset setting applies), in one launch and without a
 * barrier between the steps of different envs.  rewards (float [T][N]) / dones (uint8 [T][N]) receive every step's
 * values when not NULL; the per-env outputs of igw_buffers hold the last step's, as after igw_step_walking.
 * <- the loop of examples/run_env.py:18-26 over a recorded action sequence.  The three igw_rollout_* entry points
 * do not write the episode log: they return IGW_ERR_INVALID while igw_set_trajectory_log is enabled. */
int igw_rollout_walking_actions(igw_ctx* ctx, const int32_t* actions, int64_t T, float* rewards, uint8_t* dones,
                                void* stream);
/* ... and for the flying action space: movement float [T][N][3], camera float [T][N][2], inventory / placement
 * int32 [T][N]; exactly T calls of igw_step_flying. */
int igw_rollout_flying_actions(igw_ctx* ctx, const float* movement, const float* camera, const int32_t* inventory,
                               const int32_t* placement, int64_t T, float* rewards, uint8_t* dones, void* stream);
/* fills actions[n_steps][N] with the same counter RNG (t = t0 .. t0+n_steps-1) */
int igw_fill_actions_walking(igw_ctx* ctx, int32_t* actions, int64_t n_steps, int64_t t0, uint64_t seed,
                             int64_t env_offset, void* stream);

/* Test hook: the library's own general trig (csrc/igw_trig.h: flying mode, arbitrary poses) evaluated ON THE DEVICE for
 * n argument pairs (device pointers): sin(a), cos(a) (0 for |a| >= 2^20, outside their domain), atan2(a, b), and flags[i] bit 0 / 2 = the quick evaluation of
 * sincos / atan2 accepted, bit 1 / 3 = it accepted a value that differs from the accurate evaluation's (never set). */
int igw_debug_trig(int32_t device, int64_t n, const double* a, const double* b, double* sin_out, double* cos_out,
                   double* atan_out, uint8_t* flags, void* stream);

/* Stateless Task evaluation for n (target, grid) pairs: buffers int8 [n][IGW_GRID_STRIDE];
 * full_grid / invariant may be NULL; outputs int32: max_int[n], argmax[n][3] = (dx, dz, rot),
 * target_size[n]; any output may be NULL. */
int igw_task_eval(int32_t device, int32_t n, const int8_t* target, const int8_t* grid,
                  const int8_t* full_grid, const uint8_t* invariant, int32_t* max_int,
                  int32_t* argmax, int32_t* target_size, void* stream);

#ifdef __cplusplus
}
#endif
#endif
