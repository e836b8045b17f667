"""GeneralsVecEnv in device_outputs mode, 65,536 envs, a fixed number of steps: run under
`rocprofv3 --kernel-trace --stats` to see which of its kernels the step time goes to."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
BB = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = GeneralsVecEnv(num_envs=BB, board_width=20, board_height=20, max_players=4, device_outputs=True)
obs, info = env.reset(seed=1)
acts = torch.randint(0, 2000, (BB,), device="cuda", dtype=torch.int64)
for _ in range(60):
    env.step(acts)
torch.cuda.synchronize()
