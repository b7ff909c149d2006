"""``OnPolicyRunner`` (upstream ``rsl_rl/runners/on_policy_runner.py`` @ v2.3.1 -- third-party, absent from the
reference tree; call sites: reference scripts/reinforcement_learning/rsl_rl/train.py:167-183, play.py:115-134).

One process per GPU.  ``learn()`` = per iteration: T env steps (policy act -> env.step -> storage), GAE, PPO update.
The whole T-step rollout can be captured once into a hipGraph (``use_graph=True``) and replayed: the env kernels are
launched on torch's capturing stream, the state feed's snapshot pointers and the storage slots are baked per step.
Episode statistics stay on the device; nothing in the loop synchronises with the host unless logging asks for it.
"""

from __future__ import annotations

import json
import os
import subprocess
import time

import torch
import torch.distributed as dist

from .actor_critic import ActorCritic
from .gemm_tuning import enable_recorded_gemm_tuning
from .ppo import PPO


class ScalarWriter:
    """``add_scalar(tag, value, step)`` like the TensorBoard ``SummaryWriter`` upstream's runner logs through (tensorboard is not in
    this image): one JSON line per scalar in ``<log_dir>/scalars.jsonl``, and the last value of every tag in ``.last``."""

    def __init__(self, log_dir: str):
        os.makedirs(log_dir, exist_ok=True)
        self._f = open(os.path.join(log_dir, "scalars.jsonl"), "a")
        self.last: dict = {}

    def add_scalar(self, tag: str, value, step: int):
        v = float(value)
        self.last[tag] = v
        self._f.write(json.dumps({"tag": tag, "value": v, "step": int(step)}) + "\n")

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


class _NoFusedInference:
    ok = False

    def refresh(self):
        pass


class OnPolicyRunner:
    def __init__(self, env, train_cfg: dict, log_dir: str | None = None, device="cpu", use_graph: bool = False):
        self.cfg = train_cfg
        self.alg_cfg = dict(train_cfg["algorithm"])
        self.policy_cfg = dict(train_cfg["policy"])
        self.device = torch.device(device)
        self.env = env
        # isaaclab_rl/rsl_rl/rl_cfg.py:22,62-99,107-176: upstream's runner eval()s these names.  Only the feed-forward ActorCritic +
        # PPO pair is built here; anything else (ActorCriticRecurrent, the fork's ActorCriticCascade / PPOCA, Distillation) must not
        # silently train a plain PPO
        pol_cls = self.policy_cfg.pop("class_name", "ActorCritic")
        alg_cls = self.alg_cfg.get("class_name", "PPO")
        if pol_cls != "ActorCritic":
            raise NotImplementedError(f"policy class '{pol_cls}' is not implemented (only 'ActorCritic'); SURVEY.md section 8 scope")
        if alg_cls != "PPO":
            raise NotImplementedError(f"algorithm class '{alg_cls}' is not implemented (only 'PPO'); SURVEY.md section 8 scope")
        self._configure_multi_gpu()
        if self.device.type == "cuda":
            enable_recorded_gemm_tuning()
        obs, extras = self.env.get_observations()
        num_obs = obs.shape[1]
        num_privileged_obs = extras["observations"]["critic"].shape[1] if "critic" in extras["observations"] else num_obs
        self.privileged_obs_type = "critic" if "critic" in extras["observations"] else None
        policy = ActorCritic(num_obs, num_privileged_obs, self.env.num_actions, **self.policy_cfg).to(self.device)
        self.alg_cfg.pop("class_name", None)
        self.alg = PPO(policy, device=self.device, multi_gpu_cfg=self.multi_gpu_cfg, **self.alg_cfg)
        self.num_steps_per_env = int(train_cfg["num_steps_per_env"])
        self.save_interval = int(train_cfg.get("save_interval", 50))
        self.empirical_normalization = bool(train_cfg.get("empirical_normalization", False))
        if self.empirical_normalization:
            from .normalizer import EmpiricalNormalization

            self.obs_normalizer = EmpiricalNormalization(shape=[num_obs], until=int(1.0e8)).to(self.device)
            self.privileged_obs_normalizer = EmpiricalNormalization(shape=[num_privileged_obs], until=int(1.0e8)).to(self.device)
        else:
            self.obs_normalizer = torch.nn.Identity().to(self.device)
            self.privileged_obs_normalizer = torch.nn.Identity().to(self.device)
        # Without a privileged ("critic") observation group the critic reads the policy observations: no second (T, N, D) buffer is
        # kept -- upstream stores a copy; here the minibatch hands the SAME rows to both networks (half the gather traffic, and the two
        # first layers become one stacked GEMM).  A buffer that the fused rollout does not fill must not exist: it would feed the
        # critic zeros.
        priv_shape = [num_privileged_obs] if self.privileged_obs_type is not None else [0]
        self.alg.init_storage("rl", self.env.num_envs, self.num_steps_per_env, [num_obs], priv_shape, [self.env.num_actions])
        self.disable_logs = self.is_distributed and self.gpu_global_rank != 0
        self.log_dir = log_dir
        self.current_learning_iteration = 0
        self.tot_timesteps = 0
        self.tot_time = 0.0
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        self.alg.update_graph = self.use_graph and os.getenv("IMX_UPDATE_GRAPH", "1") != "0"
        self._graph = None
        N = self.env.num_envs
        self._cur_reward_sum = torch.zeros(N, device=self.device)
        self._cur_episode_length = torch.zeros(N, device=self.device)
        self._ep_stats = torch.zeros(3, device=self.device)  # finished episodes: sum reward, sum length, count
        # running sum of the env's extras["log"] entries over the steps of an iteration (upstream: ep_infos.append(infos["log"]) per step)
        n_log = self.env.unwrapped._log_out.numel() if hasattr(self.env.unwrapped, "_log_out") else 1
        self._log_accum = torch.zeros(n_log, device=self.device)
        self._log_steps = 0
        self._obs = obs
        # upstream learn(): privileged_obs = extras["observations"].get(privileged_obs_type, obs)
        self._privileged_obs = extras["observations"][self.privileged_obs_type] if self.privileged_obs_type is not None else obs
        self.writer = None
        self.git_status_repos: list[str] = []
        self._act_seed = int(train_cfg.get("seed", 42)) * 1000003 + self.gpu_global_rank
        self.collection_time = self.learn_time = 0.0

    # ---- distributed ---------------------------------------------------------------------------------------------
    def _configure_multi_gpu(self):
        self.gpu_world_size = int(os.getenv("WORLD_SIZE", "1"))
        self.is_distributed = self.gpu_world_size > 1 or os.getenv("IMX_FORCE_DIST") == "1"
        if not self.is_distributed:
            self.gpu_local_rank = self.gpu_global_rank = 0
            self.multi_gpu_cfg = None
            return
        self.gpu_local_rank = int(os.getenv("LOCAL_RANK", "0"))
        self.gpu_global_rank = int(os.getenv("RANK", "0"))
        self.multi_gpu_cfg = {"global_rank": self.gpu_global_rank, "local_rank": self.gpu_local_rank,
                              "world_size": self.gpu_world_size}
        # IMX_REHEARSE_ONE_GPU=1: every rank on cuda:0 (with a gloo group): the multi-rank logic on a one-GPU box, everything but RCCL
        rehearsal = os.getenv("IMX_REHEARSE_ONE_GPU") == "1"
        if (self.device.type == "cuda" and self.device.index is not None and self.device.index != self.gpu_local_rank
                and not rehearsal):
            raise ValueError(f"Device '{self.device}' does not match expected device for local rank '{self.gpu_local_rank}'.")
        if not dist.is_initialized():
            dist.init_process_group(backend="nccl" if self.device.type == "cuda" and not rehearsal else "gloo",
                                    rank=self.gpu_global_rank, world_size=self.gpu_world_size)
        if self.device.type == "cuda":
            torch.cuda.set_device(0 if rehearsal else self.gpu_local_rank)
            # the update keeps three HIP streams busy and RCCL adds its own: with ROCm's default of 4 hardware queues two of them share
            # one (+14 % update time, tools/dist_overhead.py).  The package sets GPU_MAX_HW_QUEUES=8 at import unless HIP was already
            # up or the user chose a value: say so instead of silently running slower.
            try:
                q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
            except ValueError:
                q = 4
            if q < 8 and self.gpu_global_rank == 0:
                import warnings

                warnings.warn(f"GPU_MAX_HW_QUEUES={q}: the data-parallel update wants 8 hardware queues (export it before the first HIP call, "
                              "or import isaaclab_amd before torch.cuda is touched); continuing, ~10 % slower updates", RuntimeWarning)

    # ---- rollout ---------------------------------------------------------------------------------------------------
    def _fusable(self) -> bool:
        from ..env import ManagerBasedRLEnv
        from .vecenv_wrapper import RslRlVecEnvWrapper

        return (isinstance(self.env, RslRlVecEnvWrapper) and isinstance(self.env.unwrapped, ManagerBasedRLEnv)
                and self.alg.policy.noise_std_type == "scalar"
                and self.device.type == "cuda" and not self.empirical_normalization)

    def _rollout(self):
        if self._fusable():
            return self._rollout_fused()
        obs, privileged_obs = self._obs, self._privileged_obs
        for _ in range(self.num_steps_per_env):
            actions = self.alg.act(obs, privileged_obs)
            obs, rewards, dones, infos = self.env.step(actions)
            obs = self.obs_normalizer(obs)  # upstream: normalise right after env.step (the first obs stays raw)
            if self.privileged_obs_type is not None:  # upstream on_policy_runner.py: the critic reads its own observation group
                privileged_obs = self.privileged_obs_normalizer(infos["observations"][self.privileged_obs_type])
            else:
                privileged_obs = obs
            self.alg.process_env_step(rewards, dones, infos)
            # episode book-keeping on the device (upstream pulls finished episodes to the host every step)
            self._cur_reward_sum += rewards
            self._cur_episode_length += 1
            done_f = dones.to(torch.float32)
            self._ep_stats[0] += (self._cur_reward_sum * done_f).sum()
            self._ep_stats[1] += (self._cur_episode_length * done_f).sum()
            self._ep_stats[2] += done_f.sum()
            self._cur_reward_sum *= 1.0 - done_f
            self._cur_episode_length *= 1.0 - done_f
            log_out = getattr(self.env.unwrapped, "_log_out", None)
            if log_out is not None:  # ep_infos.append(infos["log"]) per step, kept as a running sum on the device
                self._log_accum += log_out
        self._privileged_obs = privileged_obs
        if self._graph_capturing:
            self._obs_out.copy_(obs)
        return obs

    fuse_launches = True  # False: the round-2 split (imx_mlp_infer, imx_policy_act, imx_action_process, ..., imx_rollout_post: 6 launches per step)

    def _rollout_fused(self):
        """Per step THREE launches: ``imx_mlp_infer_act`` (both MLPs + PPO.act's sampling / log-prob / transition writes + the
        ActionManager's action processing in the actor head's epilogue), ``imx_terminations_rewards_rollout`` (terminations, rewards,
        resets + the wrapper's dones, the time-out bootstrap and the episode statistics into slot t) and ``imx_observations`` (+ the
        step tail and the log accumulation).  Every transition is written straight into its storage slot.  Envs with a privileged
        ("critic") observation group, Python-evaluated terms aside, take the split path (``fuse_launches = False``), which produces
        bit-identical storage contents (tests/test_kernels_gpu.py)."""
        import ctypes
        import math

        from .. import _lib
        from .._lib import ImxPolicyAct, ImxRolloutSlot, check, lib
        from .ppo import mlp_forward

        L = lib()
        alg, env, st = self.alg, self.env.unwrapped, self.alg.storage
        pol = alg.policy
        N, A, D = env.num_envs, env.plan.action_dim, env.plan.obs_dim
        stream = _lib.current_stream(self.device)
        step_ptr = env._counters[2:3].data_ptr()
        bootstrap = 0 if env.is_finite_horizon else 1
        obs = self._obs
        priv = self.privileged_obs_type  # a "critic" observation group: the critic reads it, the storage keeps it
        cobs = self._privileged_obs if priv is not None else obs
        if self._infer is not None:
            self._infer.refresh()  # the update changed the parameters: padded weight copies follow (inside the graph too)
        for t in range(self.num_steps_per_env):
            if self._infer is None:
                from .ppo import FusedInference

                # one launch for both networks needs a shared input; with a critic group the two stacks run side by side instead
                self._infer = FusedInference(alg._actor_layers, alg._critic_layers) if priv is None else _NoFusedInference()
                self._mu_buf = torch.empty(N, A, device=self.device)
                self._value_buf = torch.empty(N, 1, device=self.device)
            fused = self.fuse_launches and self._infer.ok and priv is None and obs.is_contiguous() and obs.shape[1] == D
            if fused:
                # launch 1: actor + critic + PPO.act + ActionManager.process_action
                act = ImxPolicyAct(std_d=pol.std.data_ptr(), seed=self._act_seed, step_counter_d=step_ptr,
                                   actions_out_d=st.actions[t].data_ptr(), logp_out_d=st.actions_log_prob[t].data_ptr(),
                                   mu_out_d=st.mu[t].data_ptr(), sigma_out_d=st.sigma[t].data_ptr(), obs_out_d=st.observations[t].data_ptr(),
                                   plan=env._plan_h.value, state=ctypes.pointer(env._state()), buf=ctypes.pointer(env._bufs),
                                   pre_clip=math.inf if env.clip_actions is None else float(env.clip_actions))
                self._infer(obs, None, st.values[t], act=act)
                # launch 2 (+ launch 3, the observations): the step kernel writes slot t itself; the log sum rides in the step tail
                slot = ImxRolloutSlot(value_t=st.values[t].data_ptr(), rewards_out=st.rewards[t].data_ptr(), dones_out=st.dones[t].data_ptr(),
                                      cur_reward_sum=self._cur_reward_sum.data_ptr(), cur_ep_len=self._cur_episode_length.data_ptr(),
                                      ep_stats3=self._ep_stats.data_ptr(), gamma=float(alg.gamma), bootstrap_time_outs=bootstrap)
                env._bufs.log_accum = self._log_accum.data_ptr()
                try:
                    obs_dict = env._step_after_action(slot)[0]
                finally:
                    env._bufs.log_accum = None
                obs = obs_dict["policy"]
                continue
            if self._infer.ok:  # both networks, all layers, one launch (activations stay in LDS)
                mu, value = self._mu_buf, self._value_buf
                self._infer(obs, mu, value)
            else:
                side = alg._side_stream()
                if side is not None:  # critic beside the actor (fork/join is capturable: both streams join the graph)
                    main = torch.cuda.current_stream(self.device)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        value, _ = mlp_forward(alg._critic_layers, cobs)
                    mu, _ = mlp_forward(alg._actor_layers, obs)
                    main.wait_stream(side)
                    value.record_stream(main)
                else:
                    mu, _ = mlp_forward(alg._actor_layers, obs)
                    value, _ = mlp_forward(alg._critic_layers, cobs)
            if priv is not None:
                st.privileged_observations[t].copy_(cobs)
            check(L.imx_policy_act(N, A, D, mu.data_ptr(), pol.std.data_ptr(), value.data_ptr(), obs.data_ptr(),
                                   self._act_seed, step_ptr, st.actions[t].data_ptr(), st.actions_log_prob[t].data_ptr(),
                                   st.mu[t].data_ptr(), st.sigma[t].data_ptr(), st.values[t].data_ptr(),
                                   st.observations[t].data_ptr(), None, stream))
            obs_dict, rew, terminated, truncated, _ = env.step(st.actions[t])
            check(L.imx_rollout_post(N, rew.data_ptr(), terminated.data_ptr(), truncated.data_ptr(), st.values[t].data_ptr(),
                                     float(alg.gamma), bootstrap, st.rewards[t].data_ptr(), st.dones[t].data_ptr(), None,
                                     self._cur_reward_sum.data_ptr(), self._cur_episode_length.data_ptr(),
                                     self._ep_stats.data_ptr(), env._log_out.data_ptr(), self._log_accum.data_ptr(),
                                     self._log_accum.numel(), stream))
            obs = obs_dict["policy"]
            if priv is not None:
                cobs = obs_dict[priv]
        self._privileged_obs = cobs if priv is not None else obs
        st.step = self.num_steps_per_env
        if self._graph_capturing:
            self._obs_out.copy_(obs)
            if priv is not None:
                self._priv_out.copy_(cobs)
        return obs

    _graph_capturing = False
    _infer = None

    def collect(self):
        """One rollout of ``num_steps_per_env`` env steps into the storage (eager, or one hipGraph replay)."""
        with torch.inference_mode():
            if not self.use_graph:
                self._obs = self._rollout()
                return
            if self._graph is None:
                # warm up eagerly on a side stream (allocator, lazy inits), then capture the same sequence once
                s = torch.cuda.Stream(self.device)
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    self._obs = self._rollout()
                    self.alg.storage.clear()
                torch.cuda.current_stream(self.device).wait_stream(s)
                torch.cuda.synchronize(self.device)
                self._obs_in = self._obs.clone()
                self._obs_out = torch.empty_like(self._obs)
                self._obs = self._obs_in
                if self.privileged_obs_type is not None:
                    self._priv_in = self._privileged_obs.clone()
                    self._priv_out = torch.empty_like(self._privileged_obs)
                    self._privileged_obs = self._priv_in
                self._graph = torch.cuda.CUDAGraph()
                self._graph_capturing = True
                feed_idx = self.env.unwrapped.feed.index
                with torch.cuda.graph(self._graph):
                    self._rollout()
                self._graph_capturing = False
                if self.env.unwrapped.feed.index != feed_idx:
                    raise RuntimeError("rollout graph needs num_steps_per_env to be a multiple of the feed's snapshot count")
                self._obs = self._obs_in
                if self.privileged_obs_type is not None:
                    self._privileged_obs = self._priv_in
            else:
                self._obs_in.copy_(self._obs_out)
                if self.privileged_obs_type is not None:
                    self._priv_in.copy_(self._priv_out)
            self._graph.replay()
            self.alg.storage.step = self.num_steps_per_env
            self.env.unwrapped.common_step_counter += self.num_steps_per_env

    @property
    def last_obs(self):
        return self._obs_out if self.use_graph else self._obs

    @property
    def last_critic_obs(self):
        """What upstream hands to ``compute_returns``: the privileged observations after the last step (the policy observations
        when the env has no privileged group)."""
        if self.privileged_obs_type is None:
            return self.last_obs
        return self._priv_out if self.use_graph else self._privileged_obs

    def learn(self, num_learning_iterations: int, init_at_random_ep_len: bool = False):
        if self.log_dir is not None and self.writer is None and not self.disable_logs:
            self.writer = ScalarWriter(self.log_dir)
            self._store_code_state()
        if init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf, high=int(self.env.max_episode_length))
        self.train_mode()
        if self.is_distributed:
            self.alg.broadcast_parameters()
        start_iter = self.current_learning_iteration
        tot_iter = start_iter + num_learning_iterations
        for it in range(start_iter, tot_iter):
            t0 = time.perf_counter()
            self.collect()
            with torch.inference_mode():
                self.alg.compute_returns(self.last_critic_obs)
            if self.writer is not None:
                torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None  # logging runs: honest phase times
            t1 = time.perf_counter()
            self.alg.update()
            if self.writer is not None and self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            t2 = time.perf_counter()
            self.collection_time, self.learn_time = t1 - t0, t2 - t1
            self.current_learning_iteration = it + 1
            self.tot_timesteps += self.num_steps_per_env * self.env.num_envs * self.gpu_world_size
            self.tot_time += self.collection_time + self.learn_time
            if self.writer is not None:
                self.log(it, tot_iter)
                if (it + 1) % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, f"model_{it + 1}.pt"))
        if self.writer is not None:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))
            self.writer.flush()

    def log(self, it: int, tot_iter: int, width: int = 80, pad: int = 35):
        """Upstream ``OnPolicyRunner.log``: the scalars IsaacLab's benchmark reads back (scripts/benchmarks/benchmark_rsl_rl.py:
        220-231: ``Perf/total_fps``, ``Perf/collection time``, ``Perf/learning_time``, ``Train/mean_reward``,
        ``Train/mean_episode_length``), the losses, and the per-iteration mean of the env's ``extras["log"]`` entries
        (``Episode_Reward/<term>``, ``Episode_Termination/<term>``, envs/manager_based_rl_env.py:365-389).  ONE host read per
        iteration (the sums live on the device; upstream pulls finished episodes every step).  Deviation: ``Train/mean_reward`` /
        ``mean_episode_length`` average the episodes finished in THIS iteration (upstream: a window of the last 100 episodes)."""
        env = self.env.unwrapped
        collection_size = self.num_steps_per_env * self.env.num_envs * self.gpu_world_size
        iteration_time = self.collection_time + self.learn_time
        fps = int(collection_size / max(iteration_time, 1e-9))
        w = self.writer
        index = getattr(env, "_log_index", None) or {k: i for i, k in enumerate(getattr(env, "_log_views", {}))}
        acc = self._log_accum.tolist()
        self._log_accum.zero_()
        ep_string = ""
        for key, i in index.items():  # mean over the steps of the iteration of infos["log"][key] (Episode_*, Metrics/*, Curriculum/*)
            value = acc[i] / self.num_steps_per_env
            w.add_scalar(key if "/" in key else "Episode/" + key, value, it)
            ep_string += f"""{f'Mean episode {key}:':>{pad}} {value:.4f}\n"""
        loss = self.alg.loss_dict()
        for key, value in loss.items():
            w.add_scalar(f"Loss/{key}", value, it)
        w.add_scalar("Loss/learning_rate", self.alg.learning_rate, it)
        pol = self.alg.policy
        std = pol.std if pol.noise_std_type == "scalar" else torch.exp(pol.log_std)
        mean_std = float(std.detach().mean())
        w.add_scalar("Policy/mean_noise_std", mean_std, it)
        w.add_scalar("Perf/total_fps", fps, it)
        w.add_scalar("Perf/collection time", self.collection_time, it)
        w.add_scalar("Perf/learning_time", self.learn_time, it)
        stats = self.episode_stats()
        self._ep_stats.zero_()
        if stats["episodes"] > 0:
            self._last_train = (stats["mean_reward"], stats["mean_episode_length"])
        if getattr(self, "_last_train", None) is not None:
            w.add_scalar("Train/mean_reward", self._last_train[0], it)
            w.add_scalar("Train/mean_episode_length", self._last_train[1], it)
            w.add_scalar("Train/mean_reward/time", self._last_train[0], int(self.tot_time))
            w.add_scalar("Train/mean_episode_length/time", self._last_train[1], int(self.tot_time))
        title = f" Learning iteration {it}/{tot_iter} "
        lines = [f"""{'#' * width}""", f"""{title.center(width, ' ')}""", "",
                 f"""{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {self.collection_time:.3f}s, learning {self.learn_time:.3f}s)""",
                 f"""{'Mean action noise std:':>{pad}} {mean_std:.2f}"""]
        lines += [f"""{f'Mean {k} loss:':>{pad}} {v:.4f}""" for k, v in loss.items()]
        if getattr(self, "_last_train", None) is not None:
            lines += [f"""{'Mean reward:':>{pad}} {self._last_train[0]:.2f}""", f"""{'Mean episode length:':>{pad}} {self._last_train[1]:.2f}"""]
        self.last_log_string = "\n".join(lines) + "\n" + ep_string
        if os.getenv("IMX_RUNNER_QUIET") != "1":
            print(self.last_log_string)

    def episode_stats(self) -> dict:
        s = self._ep_stats.tolist()
        n = max(s[2], 1.0)
        return {"mean_reward": s[0] / n, "mean_episode_length": s[1] / n, "episodes": s[2]}

    # ---- checkpoint / inference (train.py / play.py surface) ---------------------------------------------------
    def save(self, path: str, infos=None):
        """Upstream checkpoint layout: ``model_state_dict``, ``optimizer_state_dict`` (a torch.optim.Adam state dict), ``iter``,
        ``infos`` and, with empirical normalisation, ``obs_norm_state_dict`` / ``privileged_obs_norm_state_dict``."""
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        d = {"model_state_dict": self.alg.policy.state_dict(), "optimizer_state_dict": self.alg.optimizer_state_dict(),
             "iter": self.current_learning_iteration, "infos": infos}
        if self.empirical_normalization:
            d["obs_norm_state_dict"] = self.obs_normalizer.state_dict()
            d["privileged_obs_norm_state_dict"] = self.privileged_obs_normalizer.state_dict()
        torch.save(d, path)

    def load(self, path: str, load_optimizer: bool = True):
        d = torch.load(path, weights_only=True, map_location=self.device)
        with torch.no_grad():
            own = self.alg.policy.state_dict()
            missing = [k for k in own if k not in d["model_state_dict"]]
            if missing:
                raise KeyError(f"checkpoint lacks policy parameters {missing}")
            for k, v in d["model_state_dict"].items():
                if k in own:
                    own[k].copy_(v)  # in place: parameters stay views of the flat bucket
        if self.empirical_normalization:
            if "obs_norm_state_dict" not in d:
                raise KeyError("the runner normalises observations (empirical_normalization=True) but the checkpoint carries no "
                               "obs_norm_state_dict: its policy was trained on raw observations")
            self.obs_normalizer.load_state_dict(d["obs_norm_state_dict"])
            self.privileged_obs_normalizer.load_state_dict(d["privileged_obs_norm_state_dict"])
        if load_optimizer and "optimizer_state_dict" in d:
            self.alg.load_optimizer_state_dict(d["optimizer_state_dict"])
        self.current_learning_iteration = d.get("iter", 0)
        return d.get("infos")

    def get_inference_policy(self, device=None):
        self.eval_mode()
        if device is not None:
            self.alg.policy.to(device)
        policy = self.alg.policy.act_inference
        if self.empirical_normalization:  # upstream: policy = lambda x: act_inference(obs_normalizer(x))
            if device is not None:
                self.obs_normalizer.to(device)
            norm, act = self.obs_normalizer, self.alg.policy.act_inference
            policy = lambda x: act(norm(x))  # noqa: E731
        return policy

    def train_mode(self):
        self.alg.policy.train()
        if self.empirical_normalization:
            self.obs_normalizer.train()
            self.privileged_obs_normalizer.train()

    def eval_mode(self):
        self.alg.policy.eval()
        if self.empirical_normalization:
            self.obs_normalizer.eval()
            self.privileged_obs_normalizer.eval()

    def add_git_repo_to_log(self, repo_file_path):
        self.git_status_repos.append(repo_file_path)

    def _store_code_state(self):
        """Upstream ``store_code_state``: ``git status`` / ``git diff`` of every registered repository next to the logs."""
        for path in self.git_status_repos:
            root = path if os.path.isdir(path) else os.path.dirname(path)
            try:
                top = subprocess.run(["git", "-C", root, "rev-parse", "--show-toplevel"], capture_output=True, text=True, timeout=20)
                if top.returncode != 0:
                    continue
                top_dir = top.stdout.strip()
                out = os.path.join(self.log_dir, "git")
                os.makedirs(out, exist_ok=True)
                name = os.path.basename(top_dir) or "repo"
                with open(os.path.join(out, f"{name}.diff"), "w") as f:
                    f.write("--- git status ---\n" + subprocess.run(["git", "-C", top_dir, "status"], capture_output=True, text=True, timeout=20).stdout)
                    f.write("\n\n--- git diff ---\n" + subprocess.run(["git", "-C", top_dir, "diff"], capture_output=True, text=True, timeout=20).stdout)
            except (OSError, subprocess.SubprocessError):
                continue
