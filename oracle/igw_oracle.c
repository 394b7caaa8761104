/*
 * igw_oracle.c -- CPU ORACLE (test infrastructure, see igw_oracle.h).
 *
 * Plain-C restatement of the reference's env.step()/reset() path.  The code
 * keeps the reference's structure and floating-point operation order (all
 * binary64, no FMA contraction: build with -ffp-contract=off, never
 * -ffast-math) and cites the Python it follows as `file:line` relative to the
 * reference root.  One deliberate representational change, verified by the
 * golden vectors: the reference's dict voxel store {(x,y,z): colour}
 * (core/world.py:34, 60-71) is answered in closed form from the dense 9x11x11
 * grid that GridWorld keeps in sync with it (env.py:136-153) plus the fixed
 * 37x37 ground plane at y=-2.
 */
#include "igw_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* gridworld/utils.py:9-24, 125-132 */
#define WALKING_SPEED 5
#define FLYING_SPEED 15
#define GRAVITY 20.0
#define MAX_JUMP_HEIGHT 1.2
#define TERMINAL_VELOCITY 50
#define PLAYER_HEIGHT 2
#define WHITE (-1)
#define GREY 0
#define AGENT_PAD 0.25 /* core/world.py:9 */

/* CPython Modules/mathmodule.c: radians(x) = x * (pi/180), degrees(x) = x * (180/pi) */
static const double PY_PI = 3.141592653589793238462643383279502884;
static double py_radians(double x) { return x * (PY_PI / 180.0); }
static double py_degrees(double x) { return x * (180.0 / PY_PI); }

static igo_fn1 g_sin = sin;
static igo_fn1 g_cos = cos;
static igo_fn2 g_atan2 = atan2;

void igo_set_trig(igo_fn1 s, igo_fn1 c, igo_fn2 a) {
    g_sin = s ? s : sin;
    g_cos = c ? c : cos;
    g_atan2 = a ? a : atan2;
}

/* tasks/task.py:8-161 -- one Task object */
typedef struct {
    int8_t grids[4][IGO_CELLS]; /* target_grids: 4 rotations (task.py:40-56) */
    uint8_t adm[4][21 * 21];    /* admissible[i] as a mask over (dx+10, dz+10) (task.py:57-72) */
    int n_rot;                  /* len(self.admissible): 4, or 1 when invariant=False (task.py:59-60) */
    int target_size;            /* task.py:35 */
    int max_int;                /* task.py:42 */
    int prev_grid_size;         /* task.py:43 */
    int right_placement, wrong_placement;
} igo_task;

struct igo_env {
    igo_config cfg;
    /* Agent, core/world.py:8-29 */
    int flying;
    double strafe[2];
    double pos[3];
    double rot[2]; /* (yaw, pitch) */
    double dy;
    int time_int_steps;
    int inventory[6];
    int active_block;
    /* GridWorld, env.py:27-115 */
    int8_t grid[IGO_CELLS];
    int step_no;
    int env_max_int; /* GridWorld.max_int, written in reset() only (env.py:241) */
    double initial_position[3];
    double initial_rotation[2];
    /* tasks */
    igo_task user;             /* self._task */
    igo_task synth;            /* self._synthetic_task (env.py:227-232) */
    int8_t init[IGO_CELLS];    /* self._synthetic_init_grid (env.py:226) */
    int8_t target[IGO_CELLS];  /* self._task.target_grid */
    int have_task;
    /* SizeReward.size (env.py:316-331) */
    int size;
    /* results of the last reset/step */
    float obs_agentPos[5];
    float obs_inventory[6];
    float obs_compass;
    double reward;
    int done;
};

static inline int cell(int y, int x, int z) { return (y * IGO_GRID_X + x) * IGO_GRID_Z + z; }

/* ------------------------------------------------------------------ Task */

/* tasks/task.py:9-72 */
static void task_init(igo_task* t, const int8_t* target, const int8_t* full_grid, int invariant) {
    int8_t full[4][IGO_CELLS];
    int full_size;
    memset(t, 0, sizeof(*t));
    t->target_size = 0; /* (target_grid != 0).sum()  task.py:35 */
    for (int i = 0; i < IGO_CELLS; i++) t->target_size += target[i] != 0;
    full_size = t->target_size; /* task.py:36-38 */
    if (full_grid) {
        full_size = 0;
        for (int i = 0; i < IGO_CELLS; i++) full_size += full_grid[i] != 0;
        memcpy(full[0], full_grid, IGO_CELLS);
    }
    memcpy(t->grids[0], target, IGO_CELLS);
    /* four rotations around the vertical axis, task.py:47-56:
       target_grids[-1][:, z, 11 - x - 1] = target_grids[-2][:, x, z] */
    for (int k = 1; k < 4; k++) {
        memset(t->grids[k], 0, IGO_CELLS);
        if (full_grid) memset(full[k], 0, IGO_CELLS);
        for (int x = 0; x < IGO_GRID_X; x++)
            for (int z = 0; z < IGO_GRID_Z; z++)
                for (int y = 0; y < IGO_GRID_Y; y++) {
                    t->grids[k][cell(y, z, IGO_GRID_X - x - 1)] = t->grids[k - 1][cell(y, x, z)];
                    if (full_grid)
                        full[k][cell(y, z, IGO_GRID_X - x - 1)] = full[k - 1][cell(y, x, z)];
                }
    }
    if (!invariant) { /* task.py:59-60: admissible = [[(0,0)]] */
        t->n_rot = 1;
        t->adm[0][10 * 21 + 10] = 1;
        return;
    }
    t->n_rot = 4;
    for (int i = 0; i < 4; i++) { /* task.py:62-72 */
        const int8_t* g = full_grid ? full[i] : t->grids[i];
        for (int dx = -IGO_GRID_X + 1; dx < IGO_GRID_X; dx++)
            for (int dz = -IGO_GRID_Z + 1; dz < IGO_GRID_Z; dz++) {
                int x0 = dx > 0 ? dx : 0, x1 = IGO_GRID_X + (dx < 0 ? dx : 0);
                int z0 = dz > 0 ? dz : 0, z1 = IGO_GRID_Z + (dz < 0 ? dz : 0);
                int cnt = 0;
                for (int y = 0; y < IGO_GRID_Y; y++)
                    for (int x = x0; x < x1; x++)
                        for (int z = z0; z < z1; z++) cnt += g[cell(y, x, z)] != 0;
                if (cnt == full_size) t->adm[i][(dx + 10) * 21 + (dz + 10)] = 1;
            }
    }
}

/* tasks/task.py:138-145 */
static int task_get_intersection(const igo_task* t, const int8_t* grid, int dx, int dz, int rot) {
    /* sls_target = target_grids[rot][:, max(dx,0):11+min(dx,0), max(dz,0):11+min(dz,0)]
       sls_grid   = grid[:, max(-dx,0):11+min(-dx,0), max(-dz,0):11+min(-dz,0)] */
    int tx0 = dx > 0 ? dx : 0, tx1 = IGO_GRID_X + (dx < 0 ? dx : 0);
    int tz0 = dz > 0 ? dz : 0, tz1 = IGO_GRID_Z + (dz < 0 ? dz : 0);
    int gx0 = -dx > 0 ? -dx : 0, gz0 = -dz > 0 ? -dz : 0;
    const int8_t* tg = t->grids[rot];
    int n = 0;
    for (int y = 0; y < IGO_GRID_Y; y++)
        for (int x = tx0; x < tx1; x++) {
            const int8_t* trow = tg + cell(y, x, tz0);
            const int8_t* grow = grid + cell(y, gx0 + (x - tx0), gz0);
            for (int z = 0; z < tz1 - tz0; z++) n += (trow[z] == grow[z]) & (trow[z] != 0);
        }
    return n;
}

/* tasks/task.py:147-161 (and :121-136 for the argmax) */
static int task_maximal_intersection(const igo_task* t, const int8_t* grid, int* argmax3) {
    int max_int = 0;
    if (argmax3) argmax3[0] = argmax3[1] = argmax3[2] = 0;
    for (int i = 0; i < t->n_rot; i++)
        for (int dx = -10; dx <= 10; dx++)     /* admissible[i] was appended in this order */
            for (int dz = -10; dz <= 10; dz++) {
                if (!t->adm[i][(dx + 10) * 21 + (dz + 10)]) continue;
                int n = task_get_intersection(t, grid, dx, dz, i);
                if (n > max_int) {
                    max_int = n;
                    if (argmax3) { argmax3[0] = dx; argmax3[1] = dz; argmax3[2] = i; }
                }
            }
    return max_int;
}

/* tasks/task.py:74-86; starting grid given dense (or NULL for None) with n_start blocks */
static void task_reset(igo_task* t, const int8_t* start_dense, int n_start) {
    if (start_dense) t->max_int = task_maximal_intersection(t, start_dense, NULL);
    else t->max_int = 0;
    t->prev_grid_size = start_dense ? n_start : 0;
    t->right_placement = 0;
    t->wrong_placement = 0;
}

/* tasks/task.py:103-119 */
static void task_step_intersection(igo_task* t, const int8_t* grid, int* right, int* wrong, int* done) {
    int grid_size = 0;
    for (int i = 0; i < IGO_CELLS; i++) grid_size += grid[i] != 0;
    int wrong_placement = t->prev_grid_size - grid_size;
    int max_int = wrong_placement != 0 ? task_maximal_intersection(t, grid, NULL) : t->max_int;
    *done = max_int == t->target_size;
    t->prev_grid_size = grid_size;
    int right_placement = max_int - t->max_int;
    t->max_int = max_int;
    t->right_placement = right_placement;
    t->wrong_placement = wrong_placement;
    *right = right_placement;
    *wrong = wrong_placement;
}

void igo_task_eval(const int8_t* target, const int8_t* full_grid, int invariant, const int8_t* grid,
                   int32_t* target_size, int32_t* adm_count4, uint8_t* adm_mask, int8_t* rot,
                   int32_t* max_int, int32_t* argmax3) {
    igo_task* t = (igo_task*)malloc(sizeof(igo_task));
    task_init(t, target, full_grid, invariant);
    if (target_size) *target_size = t->target_size;
    if (adm_count4)
        for (int i = 0; i < 4; i++) {
            adm_count4[i] = 0;
            if (i < t->n_rot)
                for (int k = 0; k < 441; k++) adm_count4[i] += t->adm[i][k];
        }
    if (adm_mask) {
        memset(adm_mask, 0, 4 * 441);
        memcpy(adm_mask, t->adm, (size_t)t->n_rot * 441);
    }
    if (rot) memcpy(rot, t->grids, 4 * IGO_CELLS);
    if (grid) {
        int am[3];
        int m = task_maximal_intersection(t, grid, am);
        if (max_int) *max_int = m;
        if (argmax3) { argmax3[0] = am[0]; argmax3[1] = am[1]; argmax3[2] = am[2]; }
    }
    free(t);
}

/* ----------------------------------------------------------------- World */

/* core/world.py:57-58 */
static int build_zone(double x, double y, double z, int pad) {
    return -5 - pad <= x && x <= 5 + pad && -5 - pad <= z && z <= 5 + pad && -1 - pad <= y &&
           y < 8 + pad;
}

/* `key in self.world` / `self.world[key]`: ground plane from World._initialize
   (core/world.py:60-71: x,z in [-18,18], y=-2, WHITE inside the build zone columns else GREY)
   plus every block mirrored in the dense grid (env.py:136-153). */
static int world_lookup(const igo_env* e, int x, int y, int z, int* colour) {
    if (y == -2) {
        if (x < -18 || x > 18 || z < -18 || z > 18) return 0;
        *colour = (x >= -5 && x <= 5 && z >= -5 && z <= 5) ? WHITE : GREY;
        return 1;
    }
    if (x < -5 || x > 5 || z < -5 || z > 5 || y < -1 || y >= 8) return 0;
    int c = e->grid[cell(y + 1, x + 5, z + 5)];
    if (c == 0) return 0;
    *colour = c;
    return 1;
}

/* gridworld/utils.py:57-73: int(round(v)) -- round half to even */
static void normalize(const double* p, int* out) {
    out[0] = (int)rint(p[0]);
    out[1] = (int)rint(p[1]);
    out[2] = (int)rint(p[2]);
}

/* core/world.py:73-99.  Returns 1 when a block is hit; *have_prev tells whether
   `previous` is not None. */
static int hit_test(const igo_env* e, const double* position, const double* vector, int* block,
                    int* previous, int* have_prev) {
    const int m = 5, max_distance = 8;
    double x = position[0], y = position[1], z = position[2];
    double dx = vector[0], dy = vector[1], dz = vector[2];
    int prev[3] = {0, 0, 0}, hp = 0;
    for (int it = 0; it < max_distance * m; it++) {
        double p[3] = {x, y, z};
        int key[3], col;
        normalize(p, key);
        int differs = !hp || key[0] != prev[0] || key[1] != prev[1] || key[2] != prev[2];
        if (differs && world_lookup(e, key[0], key[1], key[2], &col)) {
            block[0] = key[0]; block[1] = key[1]; block[2] = key[2];
            previous[0] = prev[0]; previous[1] = prev[1]; previous[2] = prev[2];
            *have_prev = hp;
            return 1;
        }
        prev[0] = key[0]; prev[1] = key[1]; prev[2] = key[2];
        hp = 1;
        x = x + dx / m; y = y + dy / m; z = z + dz / m;
    }
    *have_prev = 0;
    return 0;
}

/* core/world.py:145-161 */
static void get_sight_vector(const igo_env* e, double* v) {
    double x = e->rot[0], y = e->rot[1];
    double m = g_cos(py_radians(y));
    double dy = g_sin(py_radians(y));
    double dx = g_cos(py_radians(x - 90)) * m;
    double dz = g_sin(py_radians(x - 90)) * m;
    v[0] = dx; v[1] = dy; v[2] = dz;
}

/* core/world.py:163-201 */
static void get_motion_vector(const igo_env* e, double* v) {
    double dx, dy, dz;
    if (e->strafe[0] != 0 || e->strafe[1] != 0) {
        double x = e->rot[0], y = e->rot[1];
        double strafe = py_degrees(g_atan2(e->strafe[0], e->strafe[1]));
        double y_angle = py_radians(y);
        double x_angle = py_radians(x + strafe);
        if (e->flying) {
            double m = g_cos(y_angle);
            dy = g_sin(y_angle);
            if (e->strafe[1] != 0) { dy = 0.0; m = 1; }
            if (e->strafe[0] > 0) dy *= -1;
            dx = g_cos(x_angle) * m;
            dz = g_sin(x_angle) * m;
        } else {
            dy = 0.0;
            dx = g_cos(x_angle);
            dz = g_sin(x_angle);
        }
    } else {
        dy = 0.0; dx = 0.0; dz = 0.0;
    }
    v[0] = dx; v[1] = dy; v[2] = dz;
}

/* core/world.py:264-310; FACES from gridworld/utils.py:156-163 */
static void collide(igo_env* e, const double* position, int height, double* out) {
    static const int FACES[6][3] = {{0, 1, 0}, {0, -1, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 0, 1}, {0, 0, -1}};
    const double pad = AGENT_PAD;
    double p[3] = {position[0], position[1], position[2]};
    int np[3];
    normalize(position, np);
    for (int f = 0; f < 6; f++) {
        const int* face = FACES[f];
        for (int i = 0; i < 3; i++) {
            if (!face[i]) continue;
            double d = (p[i] - np[i]) * face[i];
            if (d < pad) continue;
            for (int dy = 0; dy < height; dy++) {
                int op[3] = {np[0], np[1], np[2]}, col;
                op[1] -= dy;
                op[i] += face[i];
                if (!world_lookup(e, op[0], op[1], op[2], &col)) continue;
                p[i] -= (d - pad) * face[i];
                if (f == 0 || f == 1) e->dy = 0; /* ground or ceiling: stop falling / rising */
                break;
            }
        }
    }
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
}

/* core/world.py:222-262 */
static void update_substep(igo_env* e, double dt) {
    double speed = e->flying ? FLYING_SPEED : WALKING_SPEED;
    double d = dt * speed;
    double mv[3];
    get_motion_vector(e, mv);
    double dx = mv[0] * d, dy = mv[1] * d, dz = mv[2] * d;
    if (!e->flying) {
        e->dy -= dt * GRAVITY;
        if (e->dy < -14) e->time_int_steps = 12;
        else if (e->dy < -10) e->time_int_steps = 8;
        else if (e->dy < -5) e->time_int_steps = 4;
        else e->time_int_steps = 2;
        e->dy = e->dy > -TERMINAL_VELOCITY ? e->dy : -TERMINAL_VELOCITY; /* max(dy, -50) */
    }
    dy += e->dy * dt;
    double x = e->pos[0], y = e->pos[1], z = e->pos[2];
    double cand[3] = {x + dx, y + dy, z + dz};
    double res[3] = {x, y, z};
    if (build_zone(cand[0], cand[1], cand[2], 2)) {
        collide(e, cand, PLAYER_HEIGHT, res);
    } else if (!e->flying) {
        double c2[3] = {x, y + dy, z};
        collide(e, c2, PLAYER_HEIGHT, res);
    }
    e->pos[0] = res[0]; e->pos[1] = res[1]; e->pos[2] = res[2];
}

/* core/world.py:203-220 (sustain is always False for the env, env.py:32) */
static void update(igo_env* e, double dt) {
    int m = e->time_int_steps;
    dt = dt < 0.2 ? dt : 0.2;
    for (int i = 0; i < m; i++) update_substep(e, dt / m);
    e->strafe[0] = 0; e->strafe[1] = 0;
    if (e->flying) e->dy = 0;
}

/* env.py:136-153 callbacks keep the dense grid in sync */
static void grid_set(igo_env* e, const int* pos, int kind) {
    if (build_zone(pos[0], pos[1], pos[2], 0))
        e->grid[cell(pos[1] + 1, pos[0] + 5, pos[2] + 5)] = (int8_t)kind;
}

/* core/world.py:312-332 */
static void place_or_remove_block(igo_env* e, int remove, int place) {
    if ((place && remove) || (!place && !remove)) return;
    double vector[3];
    int block[3], previous[3], have_prev = 0;
    get_sight_vector(e, vector);
    int hit = hit_test(e, e->pos, vector, block, previous, &have_prev);
    if (place) {
        if (hit && have_prev) {
            if (e->inventory[e->active_block - 1] > 0 &&
                build_zone(previous[0], previous[1], previous[2], 0)) {
                double x = e->pos[0], y = e->pos[1], z = e->pos[2];
                y = y - (PLAYER_HEIGHT - 1) + AGENT_PAD;
                double bx = previous[0], by = previous[1], bz = previous[2];
                bx -= 0.5;
                bz -= 0.5;
                if (!(bx <= x && x <= bx + 1 && bz <= z && z <= bz + 1 &&
                      ((by <= y && y <= by + 1) || (by <= (y + 1) && (y + 1) <= by + 1)))) {
                    grid_set(e, previous, e->active_block);
                    e->inventory[e->active_block - 1] -= 1;
                }
            }
        }
    }
    if (remove && hit) {
        int texture = 0;
        world_lookup(e, block[0], block[1], block[2], &texture);
        if (texture != GREY && texture != WHITE) {
            grid_set(e, block, 0);
            e->inventory[texture - 1] += 1;
        }
    }
}

/* core/world.py:434-456 after action parsing */
static void world_step(igo_env* e, const double* strafe, double dy, int inventory /*0 = None*/,
                       const double* camera, int remove, int add) {
    if (e->cfg.select_and_place && inventory != 0) { add = 1; remove = 0; } /* :444-446 */
    /* movement, :344-356 */
    e->strafe[0] += strafe[0];
    e->strafe[1] += strafe[1];
    if (dy != 0 && e->dy == 0) e->dy = sqrt(2 * GRAVITY * MAX_JUMP_HEIGHT) * dy; /* JUMP_SPEED, utils.py:21 */
    if (e->flying && dy == 0) e->dy = 0;
    if (inventory != 0) e->active_block = inventory; /* callers guarantee 1..6 (ValueError otherwise) */
    /* move_camera, :338-342 */
    {
        double x = e->rot[0] + camera[0], y = e->rot[1] + camera[1];
        double mn = 90 < y ? 90 : y;   /* min(90, y) */
        y = -90 > mn ? -90 : mn;       /* max(-90, .) */
        e->rot[0] = x; e->rot[1] = y;
    }
    place_or_remove_block(e, remove, add);
    update(e, 1 / 20.);
    double yaw = e->rot[0];
    while (yaw > 360.) yaw -= 360.;
    while (yaw < 0.0) yaw += 360.0;
    e->rot[0] = yaw;
}

/* ------------------------------------------------------------- GridWorld */

igo_env* igo_create(const igo_config* cfg) {
    igo_env* e = (igo_env*)calloc(1, sizeof(igo_env));
    e->cfg = *cfg;
    /* Agent.__init__, core/world.py:12-29 */
    e->flying = cfg->action_space == IGO_FLYING; /* env.py:78 */
    e->time_int_steps = 2;
    for (int i = 0; i < 6; i++) e->inventory[i] = 20;
    e->active_block = 1; /* BLUE */
    return e;
}

void igo_destroy(igo_env* e) { free(e); }
int64_t igo_sizeof_env(void) { return (int64_t)sizeof(igo_env); }

void igo_set_task(igo_env* e, const int8_t* target, const int8_t* start, const int8_t* full_grid,
                  int invariant) {
    memcpy(e->target, target, IGO_CELLS);
    if (start) memcpy(e->init, start, IGO_CELLS);
    else memset(e->init, 0, IGO_CELLS); /* None is treated as [] */
    task_init(&e->user, target, full_grid, invariant);
    e->have_task = 1;
}

void igo_set_initial_pose(igo_env* e, const double* p) {
    e->initial_position[0] = p[0]; e->initial_position[1] = p[1]; e->initial_position[2] = p[2];
    e->initial_rotation[0] = p[3]; e->initial_rotation[1] = p[4];
}

static void write_obs_reset(igo_env* e) { /* env.py:247-254 */
    for (int i = 0; i < 6; i++) e->obs_inventory[i] = (float)e->inventory[i];
    e->obs_compass = 0.f;
    for (int i = 0; i < 5; i++) e->obs_agentPos[i] = 0.f;
}

/* env.py:206-261 */
void igo_reset(igo_env* e) {
    int n_start = 0;
    int8_t syn_target[IGO_CELLS];
    e->size = 0; /* SizeReward.reset, env.py:321-323 */
    e->step_no = 0;
    for (int i = 0; i < IGO_CELLS; i++) n_start += e->init[i] != 0;
    task_reset(&e->user, e->init, n_start); /* self._task.reset(), env.py:218 */
    /* synthetic task with only the diff blocks, env.py:224-232 */
    for (int i = 0; i < IGO_CELLS; i++) syn_target[i] = (int8_t)(e->target[i] - e->init[i]);
    task_init(&e->synth, syn_target, NULL, 1);
    task_reset(&e->synth, NULL, 0);
    /* remove placed blocks, add the starting ones, env.py:234-238 */
    memcpy(e->grid, e->init, IGO_CELLS);
    e->pos[0] = e->initial_position[0]; e->pos[1] = e->initial_position[1];
    e->pos[2] = e->initial_position[2];
    e->rot[0] = e->initial_rotation[0]; e->rot[1] = e->initial_rotation[1];
    e->env_max_int = task_maximal_intersection(&e->user, e->grid, NULL); /* env.py:241 */
    for (int i = 0; i < 6; i++) e->inventory[i] = 20;
    for (int i = 0; i < IGO_CELLS; i++)
        if (e->init[i] != 0) e->inventory[e->init[i] - 1] -= 1; /* env.py:244-246 */
    write_obs_reset(e);
    e->reward = 0;
    e->done = 0;
}

/* env.py:276-303 after World.step, then SizeReward.step (env.py:325-331) */
static void finish_step(igo_env* e) {
    int8_t syn[IGO_CELLS];
    int right, wrong, done;
    for (int i = 0; i < 6; i++) e->obs_inventory[i] = (float)e->inventory[i];
    e->obs_compass = (float)(e->rot[0] - 180.);
    e->obs_agentPos[0] = (float)e->pos[0];
    e->obs_agentPos[1] = (float)e->pos[1];
    e->obs_agentPos[2] = (float)e->pos[2];
    e->obs_agentPos[3] = (float)e->rot[1];
    e->obs_agentPos[4] = (float)e->rot[0];
    for (int i = 0; i < IGO_CELLS; i++) syn[i] = (int8_t)(e->grid[i] - e->init[i]);
    task_step_intersection(&e->synth, syn, &right, &wrong, &done);
    done = done || (e->step_no == e->cfg.max_steps);
    double reward;
    if (right == 0) reward = wrong * e->cfg.wrong_placement_scale;
    else reward = right * e->cfg.right_placement_scale;
    if (e->cfg.size_reward) {
        int intersection = e->env_max_int;
        int mx = intersection > e->size ? intersection : e->size;
        reward = mx - e->size;
        e->size = mx;
        reward += 0.0; /* min(GridWorld.wrong_placement * 0.02, 0) with wrong_placement == 0 forever */
    }
    e->reward = reward;
    e->done = done;
}

/* core/world.py:360-394 */
void igo_step_walking(igo_env* e, int action) {
    double strafe[2] = {0, 0}, camera[2] = {0, 0}, dy = 0;
    int inventory = 0, remove = 0, add = 0;
    if (action == 1) strafe[0] += -1;
    else if (action == 2) strafe[0] += 1;
    else if (action == 3) strafe[1] += -1;
    else if (action == 4) strafe[1] += 1;
    else if (action == 5) dy = 1;
    else if (6 <= action && action <= 11) inventory = action - 5;
    else if (action == 12) camera[0] = -5;
    else if (action == 13) camera[0] = 5;
    else if (action == 14) camera[1] = -5;
    else if (action == 15) camera[1] = 5;
    else if (action == 16) remove = 1;
    else if (action == 17) add = 1;
    e->step_no += 1;
    world_step(e, strafe, dy, inventory, camera, remove, add);
    finish_step(e);
}

/* core/world.py:396-414 (discretize=False): buttons = forward, back, left, right, jump, attack, use, hotbar */
void igo_step_walking_dict(igo_env* e, const uint8_t* b, const double* camera) {
    double strafe[2] = {0, 0};
    if (b[0]) strafe[0] += -1;
    if (b[1]) strafe[0] += 1;
    if (b[2]) strafe[1] += -1;
    if (b[3]) strafe[1] += 1;
    double jump = b[4] ? 1 : 0;   /* int(action['jump']) */
    int inventory = b[7];         /* hotbar 0 -> None */
    int remove = b[5] != 0, add = b[6] != 0;
    e->step_no += 1;
    world_step(e, strafe, jump, inventory, camera, remove, add);
    finish_step(e);
}

/* core/world.py:416-432 */
void igo_step_flying(igo_env* e, const double* movement, const double* camera, int inventory,
                     int placement) {
    double strafe[2] = {movement[0], movement[1]};
    double dy = movement[2];
    int add = placement == 1, remove = placement == 2;
    e->step_no += 1;
    world_step(e, strafe, dy, inventory, camera, remove, add);
    finish_step(e);
}

void igo_get_obs(const igo_env* e, float* agentPos, float* inventory, float* compass, int8_t* grid) {
    if (agentPos) memcpy(agentPos, e->obs_agentPos, sizeof(e->obs_agentPos));
    if (inventory) memcpy(inventory, e->obs_inventory, sizeof(e->obs_inventory));
    if (compass) *compass = e->obs_compass;
    if (grid) memcpy(grid, e->grid, IGO_CELLS);
}
double igo_get_reward(const igo_env* e) { return e->reward; }
int igo_get_done(const igo_env* e) { return e->done; }
void igo_get_internal(const igo_env* e, double* o) {
    o[0] = e->pos[0]; o[1] = e->pos[1]; o[2] = e->pos[2];
    o[3] = e->rot[0]; o[4] = e->rot[1]; o[5] = e->dy;
    o[6] = e->time_int_steps; o[7] = e->active_block;
}
void igo_get_task_state(const igo_env* e, int32_t* o) {
    o[0] = e->synth.max_int; o[1] = e->synth.prev_grid_size; o[2] = e->synth.target_size;
    o[3] = e->env_max_int; o[4] = e->step_no; o[5] = e->size;
}

/* ---------------------------------------------------------------- batch */

static void write_out(const igo_env* e, int64_t i, const igo_batch_out* out) {
    if (!out) return;
    if (out->reward) out->reward[i] = (float)e->reward;
    if (out->done) out->done[i] = (uint8_t)e->done;
}
static void write_out_obs(const igo_env* e, int64_t i, const igo_batch_out* out) {
    if (!out) return;
    igo_get_obs(e, out->agentPos ? out->agentPos + 5 * i : NULL,
                out->inventory ? out->inventory + 6 * i : NULL,
                out->compass ? out->compass + i : NULL, out->grid ? out->grid + IGO_CELLS * i : NULL);
}

/* same counter RNG as gridworld_amd/csrc (splitmix64 finaliser) */
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline int rng_action18(uint64_t seed, uint64_t env, uint64_t t) {
    uint64_t h = splitmix64(seed ^ splitmix64(env * 0x100000001B3ull + t));
    return (int)(((h >> 32) * 18ull) >> 32);
}

typedef struct {
    int kind; /* 0 walking step, 1 flying step, 2 walking rollout, 3 walking Dict step, 4 flying rollout */
    const uint8_t* buttons;
    igo_env** envs;
    int64_t lo, hi;
    const int32_t* actions;
    const float *movement, *camera;
    const int32_t *inventory, *placement;
    int autoreset;
    const igo_batch_out* out;
    int64_t T, env_offset;
    uint64_t seed;
    int64_t steps, changed;
} job_t;

static void* job_run(void* arg) {
    job_t* j = (job_t*)arg;
    for (int64_t i = j->lo; i < j->hi; i++) {
        igo_env* e = j->envs[i];
        if (j->kind == 2) {
            for (int64_t t = 0; t < j->T; t++) {
                int before = e->synth.prev_grid_size;
                igo_step_walking(e, rng_action18(j->seed, (uint64_t)(j->env_offset + i), (uint64_t)t));
                j->changed += e->synth.prev_grid_size != before;
                j->steps++;
                if (e->done && j->autoreset) igo_reset(e);
            }
            continue;
        }
        if (j->kind == 4) { /* flying rollout: movement ~ U(-1,1)^3, camera ~ U(-5,5)^2 (float32 values, as the
                             * action Box is float32), inventory U{0..6}, placement U{0..2} from the counter RNG */
            for (int64_t t = 0; t < j->T; t++) {
                uint64_t h = splitmix64(j->seed ^ splitmix64((uint64_t)(j->env_offset + i) * 0x100000001B3ull + (uint64_t)t));
                double mv[3], cam[2];
                for (int k = 0; k < 5; k++) {
                    h = splitmix64(h + (uint64_t)k);
                    float u = (float)((h >> 40) * (1.0 / 16777216.0)); /* [0, 1) */
                    if (k < 3) mv[k] = (double)(2.0f * u - 1.0f);
                    else cam[k - 3] = (double)(10.0f * u - 5.0f);
                }
                h = splitmix64(h + 5u);
                int inventory = (int)(((h >> 32) * 7ull) >> 32);
                int placement = (int)(((h & 0xffffffffull) * 3ull) >> 32);
                int before = e->synth.prev_grid_size;
                igo_step_flying(e, mv, cam, inventory, placement);
                j->changed += e->synth.prev_grid_size != before;
                j->steps++;
                if (e->done && j->autoreset) igo_reset(e);
            }
            continue;
        }
        if (j->kind == 0) {
            igo_step_walking(e, j->actions[i]);
        } else if (j->kind == 3) {
            double cam[2] = {j->camera[2 * i], j->camera[2 * i + 1]};
            igo_step_walking_dict(e, j->buttons + 8 * i, cam);
        } else {
            double mv[3] = {j->movement[3 * i], j->movement[3 * i + 1], j->movement[3 * i + 2]};
            double cam[2] = {j->camera[2 * i], j->camera[2 * i + 1]};
            igo_step_flying(e, mv, cam, j->inventory[i], j->placement[i]);
        }
        write_out(e, i, j->out);
        if (e->done && j->autoreset) igo_reset(e);
        write_out_obs(e, i, j->out);
    }
    return NULL;
}

static void run_jobs(job_t* proto, int64_t n, int nthreads, int64_t* steps, int64_t* changed) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if ((int64_t)nthreads > n) nthreads = n > 0 ? (int)n : 1;
    job_t jobs[256];
    pthread_t th[256];
    for (int k = 0; k < nthreads; k++) {
        jobs[k] = *proto;
        jobs[k].lo = n * k / nthreads;
        jobs[k].hi = n * (k + 1) / nthreads;
        jobs[k].steps = jobs[k].changed = 0;
    }
    for (int k = 1; k < nthreads; k++) pthread_create(&th[k], NULL, job_run, &jobs[k]);
    job_run(&jobs[0]);
    for (int k = 1; k < nthreads; k++) pthread_join(th[k], NULL);
    int64_t s = 0, c = 0;
    for (int k = 0; k < nthreads; k++) { s += jobs[k].steps; c += jobs[k].changed; }
    if (steps) *steps = s;
    if (changed) *changed = c;
}

void igo_batch_step_walking(igo_env** envs, int64_t n, const int32_t* actions, int autoreset,
                            int nthreads, const igo_batch_out* out) {
    job_t j;
    memset(&j, 0, sizeof(j));
    j.kind = 0; j.envs = envs; j.actions = actions; j.autoreset = autoreset; j.out = out;
    run_jobs(&j, n, nthreads, NULL, NULL);
}

void igo_batch_step_flying(igo_env** envs, int64_t n, const float* movement, const float* camera,
                           const int32_t* inventory, const int32_t* placement, int autoreset,
                           int nthreads, const igo_batch_out* out) {
    job_t j;
    memset(&j, 0, sizeof(j));
    j.kind = 1; j.envs = envs; j.movement = movement; j.camera = camera; j.inventory = inventory;
    j.placement = placement; j.autoreset = autoreset; j.out = out;
    run_jobs(&j, n, nthreads, NULL, NULL);
}

void igo_batch_step_walking_dict(igo_env** envs, int64_t n, const uint8_t* buttons, const float* camera,
                                 int autoreset, int nthreads, const igo_batch_out* out) {
    job_t j;
    memset(&j, 0, sizeof(j));
    j.kind = 3; j.envs = envs; j.buttons = buttons; j.camera = camera; j.autoreset = autoreset; j.out = out;
    run_jobs(&j, n, nthreads, NULL, NULL);
}

int64_t igo_batch_rollout_walking(igo_env** envs, int64_t n, int64_t T, uint64_t seed,
                                  int64_t env_offset, int autoreset, int nthreads,
                                  int64_t* changed_steps) {
    job_t j;
    int64_t steps = 0;
    memset(&j, 0, sizeof(j));
    j.kind = 2; j.envs = envs; j.T = T; j.seed = seed; j.env_offset = env_offset;
    j.autoreset = autoreset;
    run_jobs(&j, n, nthreads, &steps, changed_steps);
    return steps;
}

int64_t igo_batch_rollout_flying(igo_env** envs, int64_t n, int64_t T, uint64_t seed,
                                 int64_t env_offset, int autoreset, int nthreads,
                                 int64_t* changed_steps) {
    job_t j;
    int64_t steps = 0;
    memset(&j, 0, sizeof(j));
    j.kind = 4; j.envs = envs; j.T = T; j.seed = seed; j.env_offset = env_offset;
    j.autoreset = autoreset;
    run_jobs(&j, n, nthreads, &steps, changed_steps);
    return steps;
}
