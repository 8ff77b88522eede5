/*
 * mi355env.h -- C ABI of libmi355env.so, the MI355X (gfx950) lockstep vector-environment engine.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json: one call steps / resets all
 * `num_envs` sub-environments of a gymnasium.vector.VectorEnv on the GPU.  Plain pointers and sizes only;
 * no torch / numpy types.  Every entry point states the reference interface it replaces (paths relative
 * to the reference tree, gymnasium v1.4.0).
 *
 * Threading contract (same as the reference's SyncVectorEnv, which is single-threaded and not re-entrant):
 * one mi_vecenv may be used from one host thread at a time.  All work is enqueued on the env's HIP stream
 * (mi_set_stream); calls taking MI_HOST pointers synchronise that stream before returning, calls taking
 * MI_DEVICE pointers only enqueue.
 *
 * Errors: every function returns MI_OK (0) or a negative mi_status; mi_last_error() gives the message of
 * the last failure on the calling thread (the Python shim raises it as an exception, mirroring the
 * reference's AssertionError / ValueError / gymnasium.error.Error behaviour one level up).
 */
#ifndef MI355ENV_H
#define MI355ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355ENV_ABI_VERSION 7

typedef enum mi_status {
    MI_OK = 0,
    MI_ERR_INVALID_ARGUMENT = -1,
    MI_ERR_NO_DEVICE = -2,     /* no HIP device visible: the engine has NO CPU fallback */
    MI_ERR_HIP = -3,           /* a HIP runtime call failed; see mi_last_error() */
    MI_ERR_UNSUPPORTED = -4,
    MI_ERR_STATE = -5          /* e.g. step before reset/seed */
} mi_status;

/* Sub-environment dynamics.  Each value replaces the scalar env class named beside it. */
typedef enum mi_env_kind {
    MI_ENV_CARTPOLE = 0,               /* envs/classic_control/cartpole.py:119-247 (CartPole-v1)                */
    MI_ENV_PENDULUM = 1,               /* envs/classic_control/pendulum.py:102-171 (Pendulum-v1)                */
    MI_ENV_ACROBOT = 2,                /* envs/classic_control/acrobot.py:172-279,375-461 (Acrobot-v1)          */
    MI_ENV_MOUNTAIN_CAR = 3,           /* envs/classic_control/mountain_car.py:108-170 (MountainCar-v0)         */
    MI_ENV_MOUNTAIN_CAR_CONTINUOUS = 4, /* envs/classic_control/continuous_mountain_car.py:116-194              */
    /* MuJoCo family: the env class + MujocoEnv (envs/mujoco/mujoco_env.py:35-229) + the `mujoco` physics it calls  */
    MI_ENV_HALF_CHEETAH = 5,           /* envs/mujoco/half_cheetah_v5.py:153-281 + assets/half_cheetah.xml     */
    MI_ENV_ANT = 6,                    /* envs/mujoco/ant_v5.py:228-428 + assets/ant.xml                       */
    MI_ENV_HUMANOID = 7,               /* envs/mujoco/humanoid_v5.py:307-541 + assets/humanoid.xml             */
    /* ToyText: any finite MDP given as a transition table (mi_tabular_load); FrozenLake / CliffWalking / Taxi:
     * envs/toy_text/frozen_lake.py:324-348, cliffwalking.py:195-215, taxi.py:419-472, utils.py:4-8               */
    MI_ENV_TABULAR = 8,
    /* more of the MuJoCo family: same physics core, their own reward / observation glue */
    MI_ENV_HOPPER = 9,                     /* envs/mujoco/hopper_v5.py:146-343 + assets/hopper.xml                        */
    MI_ENV_WALKER2D = 10,                  /* envs/mujoco/walker2d_v5.py:151-345 + assets/walker2d_v5.xml                 */
    MI_ENV_INVERTED_PENDULUM = 11,         /* envs/mujoco/inverted_pendulum_v5.py:100-199 + assets/inverted_pendulum.xml  */
    MI_ENV_INVERTED_DOUBLE_PENDULUM = 12,  /* envs/mujoco/inverted_double_pendulum_v5.py:125-246 + its asset              */
    /* ToyText Blackjack-v1: envs/toy_text/blackjack.py:17-232 (Generator.choice card draws, dealer play-out, natural / sab rules);
     * observation row = int64[3] (player sum, dealer's showing card, usable ace); params[0] = natural, params[1] = sab */
    MI_ENV_BLACKJACK = 13,
    MI_ENV_REACHER = 14,                   /* envs/mujoco/reacher_v5.py:127-245 + assets/reacher.xml; params[0] = reward_dist_weight,
                                            * [1] = reward_control_weight, [4] = frame_skip                                  */
    MI_ENV_HUMANOID_STANDUP = 15,          /* envs/mujoco/humanoidstandup_v5.py:266-486 + assets/humanoidstandup.xml; params as HUMANOID with
                                            * [0] uph_cost_weight (unused by the reference) [5] impact_cost_weight [10],[11] impact_cost_range */
    MI_ENV_SWIMMER = 16,                   /* envs/mujoco/swimmer_v5.py:153-301 + assets/swimmer.xml (fluid forces of the medium: option density,
                                            * viscosity); params [0] forward_reward_weight [1] ctrl_cost_weight [2] reset_noise_scale
                                            * [3] exclude_current_positions [4] frame_skip                                     */
    MI_ENV_PUSHER = 17,                    /* envs/mujoco/pusher_v5.py:165-326 + assets/pusher_v5.xml; params [0] reward_near_weight [1] reward_control_weight
                                            * [4] frame_skip [5] reward_dist_weight                                            */
    MI_ENV_KIND_COUNT = 18
} mi_env_kind;

/* vector/vector_env.py:34-39 AutoresetMode; semantics of vector/sync_vector_env.py:277-319. */
typedef enum mi_autoreset_mode {
    MI_AUTORESET_NEXT_STEP = 0,
    MI_AUTORESET_SAME_STEP = 1,
    MI_AUTORESET_DISABLED = 2
} mi_autoreset_mode;

typedef enum mi_mem_location { MI_HOST = 0, MI_DEVICE = 1 } mi_mem_location;

/* Element types of the action / observation rows (mi_layout). */
typedef enum mi_dtype {
    MI_F32 = 0,
    MI_F64 = 1,
    MI_I64 = 2,
    /* action rows only (mi_step_io.actions_dtype): float64 values that were PYTHON floats on the caller's side -- the rows of a list-of-lists
     * batch, which sync_vector_env.py:274 iterate() hands to the scalar env as Python lists.  Python scalars are weak under NumPy 2's
     * promotion (NEP 50): np.float32 + float stays float32 where np.float32 + np.float64 becomes float64.  Only
     * MountainCarContinuous tells the two apart (continuous_mountain_car.py:153-155, its np.float32 state); every other kind treats the
     * value as MI_F64. */
    MI_F64_WEAK = 3
} mi_dtype;

/*
 * Construction parameters = the kwargs the reference passes to the scalar env constructor through
 * make_vec(**kwargs) (envs/registration.py:957-963) plus the TimeLimit and vectoriser settings.
 *   params[] per kind:
 *     CARTPOLE                 params[0] = sutton_barto_reward (0/1)          cartpole.py:119-121
 *     PENDULUM                 params[0] = g (default 10.0)                   pendulum.py:102
 *     MOUNTAIN_CAR(_CONTINUOUS) params[0] = goal_velocity (default 0)         mountain_car.py:108
 *     HALF_CHEETAH / ANT / HUMANOID (constructor kwargs of half_cheetah_v5.py:153-164, ant_v5.py:228-245,
 *     humanoid_v5.py:307-326; all three share slots 0-4, the last two share 5-11):
 *       [0] forward_reward_weight  [1] ctrl_cost_weight  [2] reset_noise_scale
 *       [3] exclude_current_positions_from_observation (0/1)  [4] frame_skip
 *       [5] contact_cost_weight  [6] healthy_reward  [7] terminate_when_unhealthy (0/1)  [8],[9] healthy_z_range
 *       [10],[11] ANT: contact_force_range / HUMANOID: contact_cost_range
 *     HOPPER / WALKER2D (hopper_v5.py:146-160, walker2d_v5.py:172-185): slots 0-4 and 6-9 as above,
 *       [10],[11] healthy_angle_range  [12],[13] HOPPER: healthy_state_range
 *     INVERTED_PENDULUM / INVERTED_DOUBLE_PENDULUM: [2] reset_noise_scale [4] frame_skip [6] healthy_reward (double pendulum)
 *       [12] ANT: include_cfrc_ext_in_observation / HUMANOID: include_cinert_in_observation
 *       [13],[14],[15] HUMANOID: include_cvel / include_qfrc_actuator / include_cfrc_ext _in_observation
 *     TABULAR                  params[0] = number of states, params[1] = number of actions (table: mi_tabular_load);
 *                              params[2] != 0: Taxi's fickle passenger (envs/toy_text/taxi.py:436-451,462-464) with probability params[3] -- reset
 *                              draws fickle_step, the first move with the passenger aboard re-draws the destination (Generator.choice);
 *                              layout.state_dim is then 3 (state, prob, fickle_step | buffered 32-bit half << 1)
 */
typedef struct mi_config {
    int32_t struct_size;         /* = sizeof(mi_config) */
    int32_t kind;                /* mi_env_kind */
    int32_t num_envs;            /* N >= 1 (sub-environments owned by THIS device / rank) */
    int32_t max_episode_steps;   /* TimeLimit (wrappers/common.py:116-150); <= 0 disables truncation */
    int32_t autoreset_mode;      /* mi_autoreset_mode */
    int32_t reserved[3];         /* reserved[0]: option bits (MI_CFG_*), the others 0 */
    double params[16];
} mi_config;
/* mi_config.reserved[0] bits.  MI_CFG_SOLVER_NEWTON: the MuJoCo kinds whose MJCF asks for `solver="PGS" iterations="50"` (HUMANOID,
 * HUMANOID_STANDUP: assets/humanoid.xml:8) run that solver by default; this bit selects the converged primal Newton solver instead (the
 * same convex problem solved to 1e-10 -- a deliberate, faster deviation from the reference, opt-in only). */
#define MI_CFG_SOLVER_NEWTON 1
/* MI_CFG_FAST_MATH: the classic-control kinds (CARTPOLE .. MOUNTAIN_CAR_CONTINUOUS) reproduce the reference's libm by default -- sin / cos /
 * pow(x, 2) / powf(x, 2) bit for bit, so whole trajectories are array_equal to SyncVectorEnv's (classic_control/cartpole.py:180-181,
 * pendulum.py:131, acrobot.py:263-283).  This bit selects the device's own sin / cos (<= 1 ulp away) and x * x instead: ~1.2-2x the
 * env-steps/s, results within the tolerances of tests/test_gpu_parity.py's fast-math cases.  Opt-in only. */
#define MI_CFG_FAST_MATH 2
/* MI_CFG_SHARED_RNG (MI_ENV_CARTPOLE only): the semantics of the reference's own NumPy vector environment `CartPoleVectorEnv`
 * (envs/classic_control/cartpole.py:353-505; what `make_vec("CartPole-v1")` returns by default: it is the id's vector_entry_point) instead of
 * SyncVectorEnv's.  The dynamics are the same expressions; what differs:
 *   - ONE generator for all sub-environments.  reset() draws `uniform(low, high, size=(4, N))` -- component-major: draw c * N + i is component c
 *     of sub-environment i (cartpole.py:493-500); a step re-draws the k sub-environments that finished in the previous step with
 *     `uniform(low, high, size=(4, k))` in index order (draw c * k + j for the j-th of them, :475-478).  mi_seed reads ONE generator (pcg[0..3]),
 *     mi_seed_sequence seeds it with SeedSequence(base_seed) (first_index must be 0), mi_get_rng reports it in every row.
 *   - reset bounds given to mi_reset PERSIST for the autoresets that follow (self.low / self.high, :489-491); mi_reset takes no mask.
 *   - NEXT_STEP autoreset only; with sutton_barto_reward a sub-environment that did not terminate is rewarded -0.0 (`-np.array(terminated)`, :466).
 *   The reference returns float32 rewards there (:466-468): the values (+-1, +-0) are exact in either type; mi_step_io.reward stays float64 and the host
 *   class casts.  Sub-environments do not shard across devices in this mode (draw positions depend on every other sub-environment's episode ends). */
#define MI_CFG_SHARED_RNG 4

typedef struct mi_layout {
    int32_t obs_dim;      /* observation row length (elements) */
    int32_t obs_dtype;    /* mi_dtype */
    int32_t act_dim;      /* action row length; discrete envs: 1 element of MI_I64 per env */
    int32_t act_dtype;    /* mi_dtype */
    int32_t state_dim;    /* physics state row length for mi_get_state/mi_set_state (float64) */
    int32_t info_dim;     /* per-env info row length (float64), 0 for classic control; MuJoCo columns:
                             HALF_CHEETAH: x_position, x_velocity, reward_forward, reward_ctrl   (half_cheetah_v5.py:230,241-246)
                             ANT / HUMANOID: x_position, y_position, distance_from_origin, x_velocity, y_velocity,
                                             reward_forward, reward_ctrl, reward_contact, reward_survive (ant_v5.py:359-366,384-389)
                             HUMANOID / HUMANOID_STANDUP append tendon_length[2], tendon_velocity[2] (humanoid_v5.py:486-487) */
    int32_t reserved[2];
} mi_layout;

/*
 * Buffers of one step() call, all row-major [num_envs][dim].  Pointers are all host or all device
 * (mi_mem_location).  Nullable members are skipped.
 *   actions          in   [N][act_dim] act_dtype     (i64 for Discrete -- what iterate(MultiDiscrete) yields); Box kinds: float32
 *                                               rows (layout.act_dtype, the dtype of the space and of its sampler) or, with
 *                                               actions_dtype = MI_F64, float64 rows taken UN-ROUNDED.
 *                                               NULL (ABI 7, loc == MI_DEVICE): the ON-DEVICE POLICY -- the step kernel itself draws
 *                                               `action_space.sample()` from the action stream (mi_action_seed), i.e. the call is
 *                                               `step(action_space.sample())` (utils/performance.py:82-97, the metric's own loop) in one
 *                                               launch with no host-to-device traffic; the stream advances by N * act_dim draws
 *   actions_out      out  [N][act_dim] act_dtype     with actions == NULL: the actions that were drawn (NULL to skip)
 *   obs              out  [N][obs_dim] obs_dtype
 *   reward           out  [N] f64                     (sync_vector_env.py:171)
 *   terminated       out  [N] u8 0/1                  (np.bool_ compatible)
 *   truncated        out  [N] u8 0/1
 *   final_obs        out  [N][obs_dim]  SAME_STEP only: rows of envs that finished this step (others untouched)
 *   episode_return   out  [N] f64      vector RecordEpisodeStatistics "r" (wrappers/vector/common.py:156-235):
 *   episode_length   out  [N] i32      "l"; both 0 where the env did not finish an episode this step
 *   info             out  [N][info_dim] f64   the numeric entries of the scalar env's info dict (layout.info_dim columns);
 *                                               rows of sub-envs that RESET in this call (NEXT_STEP autoreset step, SAME_STEP after a
 *                                               finished episode) hold the scalar env's reset info instead (_get_reset_info)
 *   final_info       out  [N][info_dim] f64   SAME_STEP only: the info of the step that finished the episode, for the rows of
 *                                               envs that finished this step (sync_vector_env.py:309-317 "final_info"; others untouched)
 */
typedef struct mi_step_io {
    const void *actions;
    void *obs;
    double *reward;
    uint8_t *terminated;
    uint8_t *truncated;
    void *final_obs;
    double *episode_return;
    int32_t *episode_length;
    double *info;
    double *final_info;
    /* Element type of `actions` for the Box kinds: MI_F32 (0, the default: layout.act_dtype), MI_F64 or MI_F64_WEAK.  The reference hands the caller's rows
     * to the scalar env as they are (vector/sync_vector_env.py:274 iterate(); envs/classic_control/pendulum.py:127-134 np.clip(u)[0],
     * continuous_mountain_car.py:153 action[0], envs/mujoco/mujoco_env.py:148 data.ctrl[:] = ctrl), so a float64 action array is NOT
     * rounded to float32 on the way in, and NumPy's promotion rules then make parts of the step float64 arithmetic that are float32 for a
     * float32 row (the control cost of the MuJoCo kinds, Pendulum's torque terms, MountainCarContinuous' velocity update).  MI_F64
     * reproduces that; the pinned action array of mi_host_buffers is sized for either type.  Ignored by the Discrete kinds (MI_I64). */
    int32_t actions_dtype;
    int32_t reserved;
    void *actions_out; /* ABI 7: see above (only read when actions == NULL) */
} mi_step_io;

/* Buffers of one fused rollout() call: T consecutive step()s in one launch, time-major [T][N][dim].
 * Any output pointer may be NULL (not materialised).  Device pointers only. */
typedef struct mi_rollout_io {
    const void *actions_in;   /* [T][N][act_dim] or NULL => sample on device from the action stream; element type: actions_in_dtype */
    void *actions_out;        /* [T][N][act_dim] the sampled actions (NULL to skip)                    */
    void *obs;                /* [T][N][obs_dim] */
    double *reward;           /* [T][N] */
    uint8_t *terminated;      /* [T][N] */
    uint8_t *truncated;       /* [T][N] */
    int32_t actions_in_dtype; /* Box kinds: MI_F32 (0, default) or MI_F64 rows in actions_in (see mi_step_io.actions_dtype); actions_out is always layout.act_dtype */
    int32_t reserved;
} mi_rollout_io;

/* Running totals kept on device (the multi-GPU metric all-reduce operates on these three numbers). */
typedef struct mi_stats {
    uint64_t env_steps;       /* sub-env steps that advanced dynamics (utils/performance.py:88-90 counting) */
    uint64_t reset_steps;     /* NEXT_STEP autoreset steps (not counted as env steps)                       */
    uint64_t episodes;        /* finished episodes                                                          */
    double return_sum;        /* sum of finished-episode returns                                            */
    uint64_t length_sum;      /* sum of finished-episode lengths                                            */
} mi_stats;

typedef struct mi_vecenv mi_vecenv;

/* Library / device ----------------------------------------------------------------------------------- */
int mi_abi_version(void);
const char *mi_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only host; never an error). */
int mi_device_count(void);

/* Lifetime.  Replaces SyncVectorEnv.__init__ (vector/sync_vector_env.py:76-185) / the vector_entry_point
 * creator call (envs/registration.py:963).  Fails with MI_ERR_NO_DEVICE when no GPU is present. */
int mi_create(const mi_config *cfg, int device, mi_vecenv **out);
/* Replaces VectorEnv.close_extras (vector/vector_env.py:238-240). */
void mi_destroy(mi_vecenv *env);
int mi_get_layout(const mi_vecenv *env, mi_layout *out);
/* Use an existing hipStream_t (e.g. torch's current stream) for all subsequent work.  NULL means the legacy
 * default stream (what torch.cuda.current_stream().cuda_stream is by default).  Until this is called the env
 * uses the engine's non-blocking stream of its device (one per device and process, shared by all envs).
 * Stream capture: mi_step with loc == MI_DEVICE (without a step epilogue) only launches kernels on this stream -- no synchronising HIP call,
 * no host-side state that changes from step to step -- so a caller may put the stream into capture (hipStreamBeginCapture) and record any
 * number of steps into a hipGraph (gymnasium_amd.HipVectorEnv.capture_steps does; tests/test_gpu_graph_capture.py).  The host-side checks of
 * mi_step (the sticky device error word) are made at capture time only: after replays, mi_synchronize / the next eager call raises it. */
int mi_set_stream(mi_vecenv *env, void *hip_stream);
int mi_synchronize(mi_vecenv *env);

/* Seeding.  Replaces Env.reset(seed=...) -> seeding.np_random (core.py:157-159, utils/seeding.py:10-42)
 * with the SyncVectorEnv fan-out seed+i (vector/sync_vector_env.py:207-208).
 *   mi_seed:          pcg[N][4] = {state_hi, state_lo, inc_hi, inc_lo} of each env's PCG64 (host pointer);
 *                     mask (host, nullable) selects which envs are re-seeded.
 *   mi_seed_sequence: env i <- PCG64(SeedSequence(base_seed + first_index + i)) computed on device. */
int mi_seed(mi_vecenv *env, const uint64_t *pcg, const uint8_t *mask);
int mi_seed_sequence(mi_vecenv *env, uint64_t base_seed, uint64_t first_index, const uint8_t *mask);

/* Replaces SyncVectorEnv.reset (vector/sync_vector_env.py:187-264) incl. options["reset_mask"] (:214-246)
 * and the classic-control reset options (envs/classic_control/utils.py:17-46; pendulum.py:151-162):
 *   bounds (host, nullable) = {low, high}  (Pendulum: {x_init, y_init}).
 * mask/obs are host or device pointers per `loc`; rows with mask==0 are not written. */
int mi_reset(mi_vecenv *env, const uint8_t *mask, const double *bounds, void *obs, int loc);

/* Replaces SyncVectorEnv.step (vector/sync_vector_env.py:266-337) with TimeLimit (wrappers/common.py:129-133)
 * and the scalar env's step() folded into one kernel launch. */
int mi_step(mi_vecenv *env, const mi_step_io *io, int loc);

/*
 * Asynchronous host stepping (SURVEY.md 8(b) "Threading"): replaces AsyncVectorEnv.step_async / step_wait
 * (vector/async_vector_env.py:440-521).  mi_step_async enqueues actions H2D -> step kernel -> ONE D2H of the output block and returns
 * without synchronising; mi_step_wait synchronises, reports errors and fills the arrays given to mi_step_async (host pointers).  One step
 * may be pending per env.  mi_step(io, MI_HOST) is exactly mi_step_async + mi_step_wait.
 *
 * mi_host_buffers: the env's own PINNED host arrays, laid out like the device output block (actions, obs, reward, terminated,
 * truncated, info, episode_return, episode_length, final_obs, final_info; valid until mi_destroy).  Passing these pointers to
 * mi_step / mi_step_async makes the host path zero-copy: the D2H lands where the caller reads, the actions are uploaded from where the
 * caller wrote them (SURVEY.md 8(b) "Ownership").
 */
int mi_step_async(mi_vecenv *env, const mi_step_io *io);
int mi_step_wait(mi_vecenv *env);
int mi_host_buffers(mi_vecenv *env, mi_step_io *out);

/* Transition table of a MI_ENV_TABULAR environment (host pointers, copied to the device).  Replaces the `P` dict and
 * `initial_state_distrib` the toy-text constructors build (frozen_lake.py:255-303, cliffwalking.py:118-135, taxi.py:281-334):
 *   csprob[nS][nA][K]   np.cumsum of the outcome probabilities (what categorical_sample compares against, utils.py:4-8)
 *   prob / next_state / reward / terminated [nS][nA][K]   the outcome tuples; count[nS][nA] outcomes are valid
 *   isd_csprob[nS]      np.cumsum(initial_state_distrib)
 * step: i = argmax(csprob[s][a] > rng.random()); reset: s = argmax(isd_csprob > rng.random()).  Observations are int64
 * states; layout.info_dim = 1 (the "prob" entry of the info dict).
 * num_tables > 1 (ABI 6): every array above gains a leading [num_tables] axis and env_table[N] names each sub-environment's table -- SyncVectorEnv over
 * scalar envs that were constructed differently, e.g. FrozenLake with map_name=None, where every sub-environment draws its own random map
 * (frozen_lake.py:241-242).  All tables share num_states / num_actions / max_outcomes.  num_tables 0 or 1 with env_table NULL: one table for all. */
typedef struct mi_tabular_table {
    int32_t num_states, num_actions, max_outcomes, num_tables;
    const double *csprob, *prob;
    const int32_t *next_state;
    const double *reward;
    const uint8_t *terminated;
    const int32_t *count;
    const double *isd_csprob;
    const int32_t *env_table; /* [N] host pointer: table index of each sub-environment, or NULL */
} mi_tabular_table;
int mi_tabular_load(mi_vecenv *env, const mi_tabular_table *table);

/* Random policy on device: the action space's generator (spaces/space.py:112-122 Space.seed), given as the
 * PCG64 {state_hi,state_lo,inc_hi,inc_lo}.  rollout then reproduces
 *   for t in range(T): step(action_space.sample())   (spaces/multi_discrete.py:176-178, spaces/box.py:463-465)
 * bit-for-bit in a single launch. */
int mi_action_seed(mi_vecenv *env, const uint64_t pcg[4]);
int mi_rollout(mi_vecenv *env, int T, const mi_rollout_io *io);
/* The action stream as a sampler of its own (ABI 7).  All three keep ONE position: whatever draws mi_rollout / mi_step(actions == NULL) /
 * mi_action_sample consume, the next consumer continues where the last one stopped, exactly like successive `action_space.sample()` calls on
 * the reference's seeded space (spaces/space.py:112-122 seed(), spaces/multi_discrete.py:176-178, spaces/box.py:463-465 sample()).
 *   mi_action_sample: out[T][N][act_dim] (layout.act_dtype) <- the next T batches of `action_space.sample()`; loc == MI_DEVICE only enqueues
 *                     (a host class hands them out one batch per sample() call: T launches' worth of policy in one), MI_HOST synchronises.
 *                     T == 0 just prepares the per-lane stream states (before a stream capture: mi_step(actions == NULL) is capturable).
 *   mi_action_get:    the generator that would produce the NEXT draw, as {state_hi, state_lo, inc_hi, inc_lo} (synchronises when the position
 *                     lives on the device, i.e. after mi_step(actions == NULL) / mi_action_sample): hand it to np.random.PCG64 to continue on the host.
 *   mi_action_skip:   move the position by `draws` (negative: backwards -- a host class that sampled ahead returns what it did not hand out). */
int mi_action_sample(mi_vecenv *env, int T, void *out, int loc);
int mi_action_get(mi_vecenv *env, uint64_t pcg[4]);
int mi_action_skip(mi_vecenv *env, int64_t draws);

/* Bookkeeping ------------------------------------------------------------------------------------------ */
int mi_get_stats(mi_vecenv *env, mi_stats *out);     /* synchronises */
int mi_reset_stats(mi_vecenv *env);
/* Per-env flag bits of mi_get_state/mi_set_state. */
#define MI_FLAG_NEEDS_RESET 1u /* the env finished last step (SyncVectorEnv._autoreset_envs, sync_vector_env.py:329) */
#define MI_FLAG_STATE_F32 2u   /* MountainCarContinuous: state currently held as float32 (continuous_mountain_car.py:178) */
/* MuJoCo kinds: a state row is [qpos(nq), qvel(nv), qacc_warmstart(nv), tracked_x, tracked_y] -- the Cartesian position
 * the next step's velocity reward is differenced against comes from the last forward pass and lags qpos. */
/* Physics state rows [N][state_dim] float64 (host pointers); checkpoint/resume + teacher-forced tests.
 * Replaces poking env.unwrapped.state (tests/envs/test_env_implementation.py:255-321 compares it).
 * Domain of mi_set_state for the classic-control kinds: angles that go into sin / cos within |x| < 105414336 (the range in which the engine's
 * routines ARE glibc's; CartPole and Pendulum defer to the device library beyond it, Acrobot and MountainCar -- which wrap / clip their angle --
 * assume it).  The host classes refuse rows outside it (gymnasium_amd/envs/classic_control.py _STATE_LIMITS); a binding of its own should too. */
int mi_get_state(mi_vecenv *env, double *state, int32_t *elapsed_steps, uint8_t *flags);
int mi_set_state(mi_vecenv *env, const double *state, const int32_t *elapsed_steps, const uint8_t *flags);
/* Per-env PCG64 words, same layout as mi_seed (host pointer). */
int mi_get_rng(mi_vecenv *env, uint64_t *pcg);

/*
 * Stateful vector wrappers as device epilogues of the step path.  All array arguments are DEVICE
 * pointers; work is enqueued on `hip_stream` (NULL = the legacy default stream); nothing synchronises except mi_rms_get/set.
 *
 * mi_running_stats replaces gymnasium/wrappers/utils.py:33-71 RunningMeanStd (mean[dim], var[dim], count; __init__: mean 0,
 * var 1, count = epsilon).  `dtype` is the dtype NumPy holds mean / var in there: MI_F32 for float32 observations
 * (NormalizeObservation builds RunningMeanStd(dtype=float32)), MI_F64 for float64 observations and for the scalar return
 * statistics of NormalizeReward.  mi_rms_get / mi_rms_set move the statistics to / from the host as float64 (checkpointing,
 * `wrapper.obs_rms.mean` ...).
 */
typedef struct mi_running_stats mi_running_stats;
int mi_rms_create(int device, int dim, int dtype, double epsilon, mi_running_stats **out);
void mi_rms_destroy(mi_running_stats *stats);
int mi_rms_get(mi_running_stats *stats, void *hip_stream, double *mean, double *var, double *count);       /* host pointers, synchronises */
int mi_rms_set(mi_running_stats *stats, void *hip_stream, const double *mean, const double *var, const double *count);
/*
 * NormalizeObservation.observations (gymnasium/wrappers/vector/stateful_observation.py:144-160):
 *   if update: obs_rms.update(obs)  [np.mean / np.var over axis 0, then update_mean_var_count_from_moments]
 *   out = ((obs - mean) / sqrt(var + epsilon)).astype(float32)
 * obs: [num_rows][dim] of obs_dtype (MI_F32 / MI_F64), out: [num_rows][dim] float32.
 */
int mi_normalize_observation(mi_running_stats *stats, void *hip_stream, const void *obs, int obs_dtype, int num_rows, double epsilon,
                             int update, void *out);
/*
 * NormalizeReward.step after the wrapped step (gymnasium/wrappers/vector/stateful_reward.py:150-176):
 *   active = all (same_step) | ~prev_done;  accumulated[active] = accumulated[active] * gamma * (1 - terminated) + reward
 *   if update and any(active): return_rms.update(accumulated[active]);  prev_done = terminated | truncated
 *   same_step: accumulated[prev_done] = 0;   out = reward / sqrt(return_rms.var + epsilon)
 * accumulated: [num_envs] float32 (zeroed by the caller on reset()), prev_done: [num_envs] uint8, return_rms: dim 1, MI_F64.
 */
int mi_normalize_reward(mi_running_stats *return_rms, void *hip_stream, float *accumulated, uint8_t *prev_done, const double *reward,
                        const uint8_t *terminated, const uint8_t *truncated, int num_envs, double gamma, double epsilon, int same_step,
                        int update, double *out);
/* ClipReward (gymnasium/wrappers/vector/vectorize_reward.py:115-151 -> np.clip(reward, min, max)); NULL bound = None. */
int mi_clip_reward(int device, void *hip_stream, const double *reward, int num_envs, const double *min_reward, const double *max_reward,
                   double *out);

/* The same three wrappers as the OUTPUT STAGE of the step kernel (classic-control kinds): mi_step / mi_step_async then return the wrapped
 * observations and rewards in place of the raw ones -- one extra launch per step (the normalisations need the statistics of the WHOLE batch,
 * stateful_observation.py:146-152) instead of the ten of the stand-alone passes, and no staging for host callers: the values are rewritten
 * before the step's single device-to-host copy.  Reward order: ClipReward (clip_pre) -> NormalizeReward -> ClipReward (clip_post); each part
 * optional.  The handles and arrays stay owned by the caller and must outlive the attachment; NULL detaches.  Fused rollouts (mi_rollout)
 * and resets are not affected (normalise a reset observation with mi_normalize_observation). */
typedef struct mi_step_epilogue {
    mi_running_stats *obs_rms;     /* NormalizeObservation: float32 statistics of obs_dim columns, or NULL */
    double obs_epsilon;
    int32_t obs_update;            /* update_running_mean */
    int32_t reward_update;
    mi_running_stats *return_rms;  /* NormalizeReward: scalar float64 statistics, or NULL */
    float *accumulated;            /* [N] device: discounted return per sub-environment */
    uint8_t *prev_done;            /* [N] device */
    double gamma, reward_epsilon;
    int32_t clip_pre, clip_post;   /* bit 0: has min, bit 1: has max */
    double clip_pre_min, clip_pre_max, clip_post_min, clip_post_max;
} mi_step_epilogue;
int mi_set_step_epilogue(mi_vecenv *env, const mi_step_epilogue *epilogue);

#ifdef __cplusplus
}
#endif
#endif /* MI355ENV_H */
