"""GymWrapper — host-side mirror of reference exciting_environments/gym_wrapper.py: a stateful wrapper whose
``step(action)`` returns ``(observation, reward, terminated, truncated)``. The per-step work (ODE step + reward +
terminated + truncated) is ONE fused HIP launch (``excenv_gym_step``). The reference-trajectory generator
(``update_ref`` / ``generate_new_ref``, gym_wrapper.py:170-192) has two modes: ``reset(rng_ref=<key tensor>)`` follows the
reference's per-environment key stream (state.PRNGKey -> init_state -> split -> randint, restated in ``random.py``; parity
unpinned), ``reset(rng_ref=<int | torch.Generator>)`` draws from a torch generator (own stream)."""
from __future__ import annotations

from dataclasses import fields, replace

import torch

from . import random as _random
from .registration import EnvironmentRegistry


class GymWrapper:
    def __init__(self, env, control_state=None, generate_reward=None, generate_terminated=None,
                 generate_truncated=None, ref_params=None):
        self.env = env
        if control_state is None:
            print(f"No chosen control state in the GymWrapper. Control state is set to {self.env.control_state}.")
            self.control_state = self.env.control_state
        else:
            assert type(control_state) == list, "Control state has to be a list."
            valid = [f.name for f in fields(self.env.PhysicalState)]
            for i in control_state:
                assert i in valid, f"Given control state {i} is no valid physical state {valid}."
            self.control_state = control_state
            self.env.control_state = control_state
        self.ref_gen = False
        _, init_state = self.env.vmap_reset()
        self.ref_params = ref_params or {"hold_steps_min": 10, "hold_steps_max": 1000}
        self.reference_hold_steps = torch.zeros((self.env.batch_size, 1), dtype=torch.int64, device=self.env.device)
        self.state = init_state
        # user-supplied callables (state, action/reward, env_properties) -> tensor replace the fused outputs
        self.generate_reward = generate_reward
        self.generate_truncated = generate_truncated
        self.generate_terminated = generate_terminated
        self._ref_rng = None
        self.host_hold_mirror = True  # see gym_step
        self.device_update_ref = True  # key-stream reference generator as one HIP launch (False: literal torch path)
        self._hold_min = None

    @classmethod
    def from_env(cls, env_type: EnvironmentRegistry, **env_kwargs):
        """Creates GymWrapper with environment from EnvironmentRegistry (gym_wrapper.py:61-65)."""
        return cls(env_type.make(**env_kwargs))

    def step(self, action):
        """One simulation step (gym_wrapper.py:67-86): observation [B,obs_dim], reward [B,1], terminated [B,1],
        truncated [B, n_flags]."""
        obs, reward, terminated, truncated, self.state, self.reference_hold_steps = self.gym_step(
            action, self.state, self.reference_hold_steps
        )
        return obs, reward, terminated, truncated

    def gym_step(self, action, state, reference_hold_steps):
        """gym_wrapper.py:88-130: vmap_step, then (reference generator armed) update_ref, then reward / terminated /
        truncated of the new state.

        With the generator armed a new reference is due only every hold_steps_min..max steps per environment, and drawing
        one costs a random initial state for the whole batch. The wrapper therefore mirrors the smallest hold counter on
        the host (one device read whenever references were redrawn): while it is positive no environment can be due, the
        step is the single fused launch (`excenv_gym_step`) and the counters are decremented; when it reaches zero the step
        takes the literal path. Same results either way (`test_gym_wrapper_ref_generation_fast_path_equals_literal_path`).
        The mirror is used only for the wrapper's own counter tensor (`step()`); foreign counters take the literal path."""
        env = self.env
        custom = self.generate_reward or self.generate_terminated or self.generate_truncated
        regen = len(self.control_state) and self.ref_gen
        if not custom and not regen:
            obs, reward, terminated, truncated, state = env.vmap_gym_step(state, action)
            return obs, reward, terminated, truncated, state, reference_hold_steps
        if (regen and not custom and self._ref_rng is None and _random.is_key(state.PRNGKey) and state.PRNGKey.is_cuda
                and self.device_update_ref):
            return self._gym_step_device_refgen(action, state, reference_hold_steps)
        if regen and not custom and self.host_hold_mirror and reference_hold_steps is self.reference_hold_steps:
            if self._hold_min is None:
                self._hold_min = int(reference_hold_steps.min())
            if self._hold_min > 0:  # nobody is due: update_ref would only count down
                obs, reward, terminated, truncated, state = env.vmap_gym_step(state, action)
                self._hold_min -= 1
                return obs, reward, terminated, truncated, state, reference_hold_steps - 1
            self._hold_min = None  # somebody is due: literal path below, mirror re-read on the next call
        obs, state = env.vmap_step(state, action)
        if regen:
            state, reference_hold_steps = self.update_ref(state, reference_hold_steps)
        ep = env.env_properties
        reward = (self.generate_reward or env.generate_reward)(state, action, ep)
        terminated = (self.generate_terminated or env.generate_terminated)(state, reward, ep)
        truncated = (self.generate_truncated or env.generate_truncated)(state, ep)
        return obs, reward, terminated, truncated, state, reference_hold_steps

    def _gym_step_device_refgen(self, action, state, reference_hold_steps):
        """Key-stream reference generator on the device: `excenv_update_ref` redraws the due references, new keys and hold
        counters in one launch (on copies: the inputs stay untouched), then the fused gym launch computes reward / flags
        against the NEW references while the observation columns show the OLD ones — the reference takes the observation
        before update_ref and the reward after it (gym_wrapper.py:109-126). Two launches per step instead of the literal
        path's few hundred elementwise kernels."""
        from . import _native

        env = self.env
        B = env.batch_size
        idx = [env.STATE_FIELDS.index(n) for n in self.control_state]
        old_refs = [env._t(getattr(state.reference, n), (B,)) for n in self.control_state]
        keys_in = state.PRNGKey
        if keys_in.dtype is not torch.int64 or keys_in.device != env.device or not keys_in.is_contiguous():
            keys_in = keys_in.to(device=env.device, dtype=torch.int64).contiguous()
        hold_in = reference_hold_steps
        if hold_in.dtype is not torch.int64 or hold_in.device != env.device or not hold_in.is_contiguous():
            hold_in = hold_in.to(device=env.device, dtype=torch.int64).contiguous()
        hold_in = hold_in.view(B)
        # outputs: one allocation per element type; the kernel writes every environment (redrawn or carried over)
        al = 16 // old_refs[0].element_size()
        Bp = (B + al - 1) // al * al
        new_refs = torch.empty((len(idx), Bp), dtype=env.dtype, device=env.device)[:, :B].unbind(0)
        ibuf = torch.empty(3 * B, dtype=torch.int64, device=env.device)
        keys, hold = ibuf[:2 * B].view(B, 2), ibuf[2 * B:]
        props, _keep = env._props_for(env.env_properties, B)
        _native.update_ref_to(env.ENV_ID, env.dtype, B, props, idx, old_refs, keys_in, hold_in, new_refs, keys, hold,
                              self.ref_params["hold_steps_min"], self.ref_params["hold_steps_max"])
        ref = {n: getattr(state.reference, n) for n in env.STATE_FIELDS}
        ref.update(dict(zip(self.control_state, new_refs)))
        state = replace(state, reference=env.PhysicalState(**ref), PRNGKey=keys)
        if type(action) is not torch.Tensor:
            action = torch.as_tensor(action)
        assert action.shape == (B, env.action_dim), (
            "The action needs to be of shape (batch_size, action_dim) which is "
            + f"{(B, env.action_dim)}, but {tuple(action.shape)} is given"
        )
        obs, reward, terminated, truncated, state = env._vmap_step_launch(state, action, True, obs_refs=old_refs)
        return obs, reward, terminated, truncated, state, hold.reshape(reference_hold_steps.shape)

    def reset(self, rng_env=None, rng_ref=None, initial_state=None):
        """Resets the environment to a default / random / passed initial state and (re)arms the reference generator
        when `rng_ref` is given (gym_wrapper.py:132-168)."""
        env = self.env
        if initial_state is not None:
            _, state = env.vmap_reset(initial_state=initial_state)
        else:
            _, state = env.vmap_reset(rng_env)
        if rng_ref is not None:
            if _random.is_key(rng_ref):
                # gym_wrapper.py:149-155: one key -> split(rng_ref, batch_size); [B, 2] keys are taken as they are. The
                # keys live in state.PRNGKey and advance with every new reference (generate_new_ref).
                keys = rng_ref.to(env.device)
                if keys.ndim == 1:
                    keys = _random.split(keys, env.batch_size)
                assert keys.shape[0] == env.batch_size
                state = replace(state, PRNGKey=keys)
                self._ref_rng = None
            else:
                gen = rng_ref
                if not isinstance(rng_ref, torch.Generator):
                    gen = torch.Generator(device=env.device)
                    gen.manual_seed(int(rng_ref))
                self._ref_rng = gen
            self.ref_gen = True
            need = torch.ones((env.batch_size, 1), dtype=torch.bool, device=env.device)
            state, self.reference_hold_steps = self.generate_new_ref(state, need, self.reference_hold_steps)
        else:
            self.ref_gen = False
            print("Since no PRNGKey for reference was provided, reference generation is deactivated.")
        self.state = state
        self._hold_min = None
        return env.generate_observation(state, env.env_properties), {}

    def update_ref(self, state, hold_steps):
        """gym_wrapper.py:170-175: draw a new reference where the hold counter reached zero, then count down. Key mode is a
        pure function of each environment's key. Generator mode shares ONE torch stream across the batch, so it draws only
        at steps where some environment is due (that keeps the stream independent of how the step was executed)."""
        due = hold_steps == 0
        if self._ref_rng is not None and not bool(due.any()):
            return state, hold_steps - 1
        state, hold_steps = self.generate_new_ref(state, due, hold_steps)
        return state, hold_steps - 1

    def generate_new_ref(self, state, mask, hold_steps):
        """gym_wrapper.py:177-192 for the environments selected by `mask` [B,1]: reference := the controlled fields of a random
        initial state; hold ~ randint(hold_steps_min, hold_steps_max). Key mode (state.PRNGKey holds [B,2] keys):
        init = init_state(PRNGKey); key, subkey = split(init.PRNGKey); hold = randint(subkey, (1,), min, max); PRNGKey = key."""
        env = self.env
        m = mask[:, 0]
        ref = {n: getattr(state.reference, n) for n in env.STATE_FIELDS}
        lo, hi = self.ref_params["hold_steps_min"], self.ref_params["hold_steps_max"]
        if self._ref_rng is None and _random.is_key(state.PRNGKey):
            init = env.vmap_init_state(state.PRNGKey)
            sp = _random.split(init.PRNGKey)
            new_hold = _random.randint(sp[:, 1, :], 1, lo, hi, x64=env.dtype is torch.float64).to(hold_steps.dtype)  # x64 <=> float64 arrays
            state = replace(state, PRNGKey=torch.where(mask, sp[:, 0, :], state.PRNGKey.to(sp.device)))
        else:
            init = env.vmap_init_state(self._ref_rng)
            new_hold = torch.randint(lo, hi, (env.batch_size, 1), generator=self._ref_rng, device=env.device)
        for name in self.control_state:
            ref[name] = torch.where(m, getattr(init.physical_state, name), ref[name])
        return replace(state, reference=env.PhysicalState(**ref)), torch.where(mask, new_hold, hold_steps)

    def render(self, *_, **__):
        raise NotImplementedError("To be implemented!")

    def close(self):
        raise NotImplementedError("To be implemented!")
