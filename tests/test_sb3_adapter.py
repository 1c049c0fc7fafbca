"""SB3 VecEnv adapter against a stub engine on the CPU: packaging of dones / infos / terminal observations."""
import numpy as np
import torch

from mycobotgym_amd.sb3_adapter import MyCobotSB3VecEnv
from mycobotgym_amd.spaces import Box, Dict


class _StubEngine:
    num_envs = 3
    single_action_space = Box(-1, 1, (7,), np.float32)
    single_observation_space = Dict({"observation": Box(-np.inf, np.inf, (10,), np.float64),
                                     "achieved_goal": Box(-np.inf, np.inf, (3,), np.float64),
                                     "desired_goal": Box(-np.inf, np.inf, (3,), np.float64)})

    def __init__(self):
        self.t = 0; self.seeds = []

    def _obs(self, v):
        return {"observation": torch.full((3, 10), float(v), dtype=torch.float64),
                "achieved_goal": torch.zeros(3, 3, dtype=torch.float64), "desired_goal": torch.ones(3, 3, dtype=torch.float64)}

    def reset(self, seed=None):
        self.seeds.append(seed); self.t = 0
        return self._obs(0), {}

    def step(self, actions):
        assert actions.shape == (3, 7) and actions.dtype == np.float32
        self.t += 1
        term = torch.tensor([False, self.t == 2, False]); trunc = torch.tensor([False, self.t == 2, self.t == 3])
        info = {"is_success": term.clone(), "final_observation": self._obs(100 + self.t), "_final_observation": trunc,
                "episode": {"r": torch.tensor([-1.0, -2.0, -3.0], dtype=torch.float64), "l": torch.tensor([1, 2, 3], dtype=torch.int32)}}
        return self._obs(self.t), torch.tensor([-0.5, -0.25, -1.0], dtype=torch.float64), term, trunc, info

    def compute_reward(self, ag, dg, info):
        return -torch.linalg.norm(torch.as_tensor(ag) - torch.as_tensor(dg), dim=-1)

    def close(self):
        pass


def test_vecenv_protocol():
    eng = _StubEngine(); v = MyCobotSB3VecEnv(eng)
    assert v.num_envs == 3 and v.seed(5) == [5, 6, 7]
    obs = v.reset()
    assert eng.seeds == [5] and isinstance(obs["observation"], np.ndarray) and obs["observation"].shape == (3, 10)
    obs, rew, dones, infos = v.step(np.zeros((3, 7)))
    assert rew.dtype == np.float32 and dones.dtype == bool and not dones.any() and len(infos) == 3
    assert all("terminal_observation" not in i for i in infos)
    obs, rew, dones, infos = v.step(np.zeros((3, 7)))
    assert dones.tolist() == [False, True, False]
    assert infos[1]["is_success"] and infos[1]["TimeLimit.truncated"] is False            # success, not a time-out
    assert infos[1]["terminal_observation"]["observation"][0] == 102.0 and infos[1]["episode"]["l"] == 2
    obs, rew, dones, infos = v.step(np.zeros((3, 7)))
    assert dones.tolist() == [False, False, True] and infos[2]["TimeLimit.truncated"] is True
    r = v.env_method("compute_reward", np.zeros((5, 3)), np.ones((5, 3)), None, indices=[0])
    assert len(r) == 1 and np.allclose(r[0], -np.sqrt(3))
    assert v.get_attr("num_envs", indices=0) == [3] and v.env_is_wrapped(type("Monitor", (), {})) == [True] * 3
