"""TEST INFRASTRUCTURE: empty stand-in so that gymnasium/envs/mujoco/mujoco_rendering.py imports (nothing renders in the fixture pipeline)."""
