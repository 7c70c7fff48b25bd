"""CPU tests of the host-side logic (no GPU): seed hierarchy RNG, trace driver, target formulas,
_findmax_skipnan mirror, argument errors of the public API."""
import numpy as np
import pytest

from oracle import pf_oracle as po


def test_host_philox_matches_oracle_and_kat():
    from pfmi import hostrng
    out = hostrng.philox4x32_10(np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], dtype=np.uint32),
                                (0xA4093822, 0x299F31D0))
    assert [int(v) for v in out[0]] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    for seed, t, stream in [(1, 0, 1), (2**63 + 12345, 2**33 + 7, 2), (987654321987654321, 99, 9)]:
        assert int(hostrng.rand_u64(seed, [t], stream)[0]) == po.rand_u64(seed, t, stream)
    r = hostrng.HostRNG(5)
    a = r.rand_u64(4)
    r2 = hostrng.HostRNG(5)
    assert np.array_equal(np.concatenate([r2.rand_u64(1), r2.rand_u64(3)]), a)       # counter based
    c = r.copy()
    assert np.array_equal(c.rand_u64(3), r.rand_u64(3))
    z = hostrng.HostRNG(1).randn(20001)
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05


def test_targets_match_oracle_and_gradients():
    import pfmi
    from helpers import oracle_target
    rng = np.random.default_rng(0)
    for tg in (pfmi.t_iso(7), pfmi.t_diag(9), pfmi.t_lowrank(12, r=3), pfmi.t_funnel(6)):
        X = rng.normal(size=(tg.d, 5))
        np.testing.assert_allclose(tg.logp(X), oracle_target(tg).logp(X), rtol=1e-12, atol=1e-12)
        x = rng.normal(size=tg.d)
        g = tg.grad(x)
        for i in range(tg.d):
            h = 1e-6
            xp, xm = x.copy(), x.copy(); xp[i] += h; xm[i] -= h
            assert abs((tg.logp(xp) - tg.logp(xm)) / (2 * h) - g[i]) < 1e-5 * (1 + abs(g[i]))
    cb = pfmi.CallbackTarget(3, lambda x: -0.5 * float(x @ x))
    np.testing.assert_allclose(cb.grad(np.array([1.0, -2.0, 0.5])), [-1.0, 2.0, -0.5], atol=1e-6)


def test_trace_driver_converges_and_records_log_density_gradients():
    import pfmi
    tg = pfmi.t_lowrank(40, r=4, seed=3)
    tr = pfmi.optimize_with_trace(tg, pfmi.HostRNG(1).rand(40) * 4 - 2)
    L = len(tr) - 1
    assert L > 3 and np.max(np.abs(tr.gradients[-1])) <= 1e-8
    assert tr.points.shape == (L + 1, 40) and tr.gradients.shape == (L + 1, 40)
    np.testing.assert_allclose(tr.log_densities, [float(tg.logp(p)) for p in tr.points], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(tr.gradients[2], tg.grad(tr.points[2]), rtol=1e-12, atol=1e-12)
    assert np.all(np.diff(tr.log_densities) > -1e-12)          # monotone ascent of logp
    # every step of an L-BFGS run on a convex quadratic has positive curvature -> no rejected updates
    _, hl, _, rej = po.lbfgs_history(tr.points, tr.gradients, 6)
    assert rej == 0 and hl.max() == min(6, L)


def test_findmax_mirror_and_api_errors():
    import pfmi
    from pfmi.api import _findmax_skipnan
    assert _findmax_skipnan([np.nan, 3.0, 1.0]) == (3.0, 2)
    v, i = _findmax_skipnan([np.nan, np.nan])
    assert np.isnan(v) and i == 1
    assert _findmax_skipnan([2.0, np.nan, 4.0]) == (4.0, 3)
    for xs in ([1.0, 5.0, 5.0], [np.nan, 2.0, np.nan, 2.0]):
        assert _findmax_skipnan(xs)[1] == po.findmax_skipnan(xs)[1]
    with pytest.raises(ValueError):
        pfmi.UniformSampler(0)                                   # DomainError, reference src/singlepath.jl:335
    with pytest.raises(ValueError):
        pfmi.multipathfinder(pfmi.t_iso(3), 10)                  # ArgumentError, reference src/multipath.jl:148-150
    assert pfmi.maximize_elbo(pfmi.HostRNG(0), pfmi.t_iso(3), [], 10) == (0, [])   # reference src/elbo.jl:7
    assert pfmi.DEFAULT_HISTORY_LENGTH == 6 and pfmi.DEFAULT_NDRAWS_ELBO == 5


def test_predrawn_fit_seeds_equal_the_sequential_draws():
    """api._run_paths draws the per-fit seeds of a run for the LONGEST possible trace while the device still optimises, from a copy of the
    run's rng, and advances the real rng by L_k afterwards: the seeds, the value a failed run would draw next
    (rand(rng, fit_distribution, n), src/singlepath.jl:231-233) and the rng's final state must be those of the reference's order of
    operations -- rand!(rng_k, UInt64[L_k]) (src/elbo.jl:2) after the optimisation."""
    import numpy as np
    from pfmi.hostrng import HostRNG, rand_u64_multi
    rngs = [HostRNG(s) for s in (3, 4, 5)]
    for r in rngs:
        r.rand(7)                                                  # the init sampler already consumed something
    ref = [r.copy() for r in rngs]
    Ls = [5, 0, 12]
    cap = 21
    pre = rand_u64_multi([r.copy() for r in rngs], [cap] * 3)
    for r, L, p, q in zip(rngs, Ls, pre, ref):
        seeds = p[:L]
        fail = p[L]
        r.counter += L
        np.testing.assert_array_equal(seeds, q.rand_u64(L))
        assert fail == q.copy().rand_u64(1)[0]
        assert r.counter == q.counter and r.rand_u64(1)[0] == q.rand_u64(1)[0]


def test_contiguous_blocks_of_runs_per_engine():
    import pytest
    from pfmi.api import _blocks
    assert _blocks(64, 8) == [(8 * g, 8 * g + 8) for g in range(8)] and _blocks(5, 1) == [(0, 5)]
    # uneven shards (reference: any nruns, src/multipath.jl:131-146): the first K % G engines take one more run, blocks stay contiguous
    assert _blocks(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)] and _blocks(20, 8)[3:6] == [(9, 12), (12, 14), (14, 16)]
    with pytest.raises(ValueError, match="at least one run"):
        _blocks(3, 4)


def test_host_driver_records_the_nonfinite_iterate_then_stops():
    """reference src/optimize.jl:96-105: the callback pushes (x, fx, grad) and THEN returns true for NaN / +Inf log density or a
    non-finite gradient -- the offending iterate is the last point of the trace"""
    import numpy as np
    from pfmi.optimize import optimize_with_trace

    class T:
        def logp_and_grad(self, x):
            g = -x.copy()
            if np.max(np.abs(x)) < 0.5:
                g[0] = np.nan                      # gradient breaks near the optimum
            return float(-0.5 * (x @ x)), g

    tr = optimize_with_trace(T(), np.array([3.0, -2.0, 1.5]), 6, 100)
    assert len(tr) >= 2
    assert np.all(np.isfinite(tr.gradients[:-1])) and np.all(np.isfinite(tr.points))
    assert not np.all(np.isfinite(tr.gradients[-1]))
    assert np.isfinite(tr.log_densities[-1])

    class U(T):
        def logp_and_grad(self, x):
            return (float("nan") if abs(x[0]) < 1.0 else float(-0.5 * (x @ x))), -x.copy()

    tr = optimize_with_trace(U(), np.array([3.0, -2.0, 1.5]), 6, 100)
    assert np.isnan(tr.log_densities[-1]) and np.all(np.isfinite(tr.log_densities[:-1]))


def test_uniform_sampler_and_argument_errors_like_reference_testsets():
    """reference test/singlepath.jl:139-153 (UniformSampler), :166-171 and test/multipath.jl:100-105 (ArgumentErrors)"""
    import numpy as np
    import pfmi
    for bad in (-1.0, 0.0):
        with pytest.raises(ValueError):                          # DomainError
            pfmi.UniformSampler(bad)
    for scale in (1, 2):
        for seed in (42, 38):
            sampler = pfmi.UniformSampler(scale)
            x = sampler(pfmi.HostRNG(seed), np.zeros(100))
            assert np.all((-scale <= x) & (x <= scale))
            x2 = pfmi.HostRNG(seed).rand(100) * 2 * scale - scale       # x2 .= rand.(rng) .* 2scale .- scale
            np.testing.assert_array_equal(x2, x)
    with pytest.raises(ValueError):                              # pathfinder(logp): neither dim nor init
        pfmi.pathfinder(pfmi.CallbackTarget(0, lambda x: 0.0))
    with pytest.raises(ValueError):                              # multipathfinder(l, 10; nruns = 0)
        pfmi.multipathfinder(pfmi.t_iso(5), 10, nruns=0)


def test_rand_u64_multi_one_native_call_equals_the_per_generator_calls():
    import numpy as np
    import pfmi
    from pfmi.hostrng import HostRNG, rand_u64_multi
    a = [HostRNG(s) for s in (11, 12, 13, 14)]
    b = [HostRNG(s) for s in (11, 12, 13, 14)]
    for r in a + b:
        r.counter = 9
    x = rand_u64_multi(a, [33] * 4)                      # equal counts: pfmi_host_rand_u64_multi
    y = [r.rand_u64(33) for r in b]
    assert all(np.array_equal(p, q) for p, q in zip(x, y)) and [r.counter for r in a] == [42] * 4
    x = rand_u64_multi(a, [3, 0, 5, 1])                  # ragged: one call per generator
    y = [r.rand_u64(n) for r, n in zip(b, [3, 0, 5, 1])]
    assert all(np.array_equal(p, q) for p, q in zip(x, y))
