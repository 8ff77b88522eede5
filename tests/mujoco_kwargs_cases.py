"""Non-default constructor kwargs of the eleven MuJoCo v5 classes, shared by the CPU glue test (reference env classes vs the oracle-backed engine,
tests/test_mujoco_fixture_pipeline.py) and the GPU test (HIP engine vs oracle, tests/test_gpu_mujoco.py)."""
# non-default constructor kwargs of every v5 class (reward weights, ranges, observation switches, frame_skip, noise scales): make_vec forwards them
# verbatim to the scalar envs on the reference's side and to the engine's creator on ours
CASES = {
    "HalfCheetah-v5": [dict(forward_reward_weight=2.0, ctrl_cost_weight=0.3, reset_noise_scale=0.2, exclude_current_positions_from_observation=False), dict(frame_skip=3)],
    "Ant-v5": [dict(forward_reward_weight=1.5, ctrl_cost_weight=0.7, contact_cost_weight=1e-3, healthy_reward=0.5, terminate_when_unhealthy=False, healthy_z_range=(0.3, 0.9),
                    contact_force_range=(-0.5, 0.5), reset_noise_scale=0.2, exclude_current_positions_from_observation=False, include_cfrc_ext_in_observation=False),
               dict(frame_skip=2, healthy_z_range=(0.5, 0.8))],
    "Humanoid-v5": [dict(forward_reward_weight=0.5, ctrl_cost_weight=0.2, contact_cost_weight=1e-6, contact_cost_range=(-1.0, 3.0), healthy_reward=2.0, terminate_when_unhealthy=False,
                         healthy_z_range=(1.1, 1.6), reset_noise_scale=0.05, exclude_current_positions_from_observation=False, include_cinert_in_observation=False,
                         include_cvel_in_observation=False), dict(include_qfrc_actuator_in_observation=False, include_cfrc_ext_in_observation=False, frame_skip=2)],
    "HumanoidStandup-v5": [dict(uph_cost_weight=2.0, ctrl_cost_weight=0.3, impact_cost_weight=1e-6, impact_cost_range=(-1.0, 4.0), reset_noise_scale=0.03,
                                exclude_current_positions_from_observation=False, include_cvel_in_observation=False, include_cfrc_ext_in_observation=False)],
    "Hopper-v5": [dict(forward_reward_weight=2.0, ctrl_cost_weight=0.01, healthy_reward=0.3, terminate_when_unhealthy=False, healthy_state_range=(-50.0, 50.0), healthy_z_range=(0.8, 1.4),
                       healthy_angle_range=(-0.1, 0.1), reset_noise_scale=0.01, exclude_current_positions_from_observation=False), dict(healthy_z_range=(1.0, 1.3), frame_skip=2)],
    "Walker2d-v5": [dict(forward_reward_weight=0.7, ctrl_cost_weight=0.02, healthy_reward=2.0, terminate_when_unhealthy=False, healthy_z_range=(0.9, 1.5), healthy_angle_range=(-0.5, 0.5),
                         reset_noise_scale=0.02, exclude_current_positions_from_observation=False)],
    "Swimmer-v5": [dict(forward_reward_weight=3.0, ctrl_cost_weight=0.01, reset_noise_scale=0.3, exclude_current_positions_from_observation=False)],
    "Reacher-v5": [dict(reward_dist_weight=2.0, reward_control_weight=0.3, frame_skip=3)],
    "Pusher-v5": [dict(reward_near_weight=1.5, reward_dist_weight=0.4, reward_control_weight=0.7)],
    "InvertedPendulum-v5": [dict(reset_noise_scale=0.1, frame_skip=3)],
    "InvertedDoublePendulum-v5": [dict(healthy_reward=3.0, reset_noise_scale=0.2)],
}
