"""The vmap_step host fast path (pooled output slots, cached pointer arrays) and the in-place Stepper (eager and HIP graph)
must give the bits of plain per-call stepping, keep the functional contract of vmap_step (fresh outputs, inputs untouched)
and survive the ways a caller can break the cache (mutated state leaves, foreign states, changed control_state)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import NP_DTYPE, make_env, random_state, spec_of, to_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_name", ["pmsm", "pendulum", "fluid_tank"])
def test_vmap_step_outputs_are_fresh_and_inputs_untouched(env_name):
    B, K = 1000, 70  # 70 calls: more than one slot allocation (32 slots per allocation at this size)
    env, props, keep, spec = make_env(env_name, B, torch.float32)
    st = random_state(env_name, B, np.float32, spec, seed=401)
    acts = np.random.default_rng(402).uniform(-1, 1, (K, B, env.action_dim)).astype(np.float32)
    a_dev = torch.as_tensor(acts, device=env.device)
    state0 = state = to_state(env, st)
    kept_obs, kept_states = [], []
    for k in range(K):
        obs, state = env.vmap_step(state, a_dev[k])
        kept_obs.append(obs)
        kept_states.append(state)
    torch.cuda.synchronize()
    # every returned tensor is its own memory: nothing was overwritten by later calls
    ptrs = {o.data_ptr() for o in kept_obs} | {getattr(s.physical_state, n).data_ptr() for s in kept_states for n in env.STATE_FIELDS}
    assert len(ptrs) == K * (1 + len(env.STATE_FIELDS))
    s_np = st
    for k in range(K):
        o_ref, s_np = oracle.step(env_name, "euler", s_np, acts[k], props, spec["tau"])
        if env_name == "fluid_tank":
            assert np.array_equal(kept_obs[k].cpu().numpy(), o_ref), k
        elif k < 8:
            assert np.allclose(kept_obs[k].cpu().numpy(), o_ref, rtol=1e-5, atol=1e-5), k
    for j, n in enumerate(env.STATE_FIELDS):  # the input state of the first call was never written
        assert torch.equal(getattr(state0.physical_state, n), torch.as_tensor(st[j], device=env.device))


def test_vmap_step_cache_survives_mutation_and_foreign_states():
    B = 512
    env, props, keep, spec = make_env("pendulum", B, torch.float32)
    st = random_state("pendulum", B, np.float32, spec, seed=411)
    a = torch.as_tensor(np.random.default_rng(412).uniform(-1, 1, (B, 1)).astype(np.float32), device=env.device)
    s0 = to_state(env, st)
    o1, s1 = env.vmap_step(s0, a)
    o2, s2 = env.vmap_step(s1, a)             # cache hit
    o2b, s2b = env.vmap_step(s1, a)           # s1 again (not the last returned state): miss path, same answer
    assert torch.equal(o2, o2b)
    s1.physical_state.omega = s1.physical_state.omega * 0 + 1.5   # caller swaps a leaf of a state we returned
    o3, _ = env.vmap_step(s1, a)
    fresh = to_state(env, [s1.physical_state.theta.cpu().numpy(), np.full(B, 1.5, dtype=np.float32)])
    o3_ref, _ = env.vmap_step(fresh, a)
    assert torch.equal(o3, o3_ref) and not torch.equal(o3, o2)
    # float64 host arrays as action / numpy leaves are converted, not rejected
    o4, _ = env.vmap_step(fresh, a.cpu().numpy().astype(np.float64))
    assert torch.equal(o4, o3_ref)
    # changing control_state between calls changes the observation width and the control block
    env.control_state = ["theta"]
    fresh.reference.theta = torch.full((B,), 0.5, device=env.device)
    o5, s5 = env.vmap_step(fresh, a)
    assert o5.shape == (B, 3) and torch.equal(o5[:, :2], o3_ref)
    s5.reference.theta = torch.full((B,), -0.25, device=env.device)
    o6, _ = env.vmap_step(s5, a)
    assert torch.allclose(o6[:, 2], torch.full((B,), -0.25 / np.pi, device=env.device))


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("env_name,gym", [("pmsm", False), ("cartpole", True), ("mass_spring_damper", False)])
def test_stepper_is_bit_identical_to_eager_vmap_step(env_name, gym, graph):
    B, n, rounds = 2048, 8, 3
    control = ["theta"] if env_name == "cartpole" else None
    env, props, keep, spec = make_env(env_name, B, torch.float32, control_state=control)
    st = random_state(env_name, B, np.float32, spec, seed=421)
    refs = {"theta": np.random.default_rng(5).uniform(-1, 1, B).astype(np.float32)} if control else None
    s_ref = to_state(env, st, reference=refs)
    stepper = env.make_stepper(n_steps=n, graph=graph, gym=gym)
    stepper.reset(to_state(env, st, reference=refs))
    rng = np.random.default_rng(422)
    for r in range(rounds):
        acts = torch.as_tensor(rng.uniform(-1, 1, (n, B, env.action_dim)).astype(np.float32), device=env.device)
        stepper.actions.copy_(acts)
        out = stepper.run()
        for k in range(n):
            if gym:
                o, rew, term, trunc, s_ref = env.vmap_gym_step(s_ref, acts[k])
                assert torch.equal(out[1][k], rew) and torch.equal(out[2][k], term) and torch.equal(out[3][k], trunc), (r, k)
            else:
                o, s_ref = env.vmap_step(s_ref, acts[k])
            assert torch.equal(out[0][k], o), (r, k)
        for name in env.STATE_FIELDS:
            assert torch.equal(getattr(stepper.state.physical_state, name), getattr(s_ref.physical_state, name)), (r, name)
