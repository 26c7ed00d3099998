"""Vectors for the running-sum tests (CPU harness and GPU kernels) and the serial loop they must reproduce."""
import numpy as np


def serial_sums(v):
    out = np.empty(v.size, dtype=np.float64)
    acc = np.float64(0.0)
    for i, x in enumerate(v.astype(np.float64)):
        acc = acc + x                       # one fp64 rounding per step, like `cumProb += x[i]`
        out[i] = acc
    return out


def adversarial():
    rng = np.random.default_rng(12345)
    cases = []
    lg = rng.normal(0, 3, 40000).astype(np.float32)
    e = np.exp(lg - lg.max()).astype(np.float32)
    cases += [e, (e / e.sum()).astype(np.float32), np.sort(e)[::-1].copy(), np.sort(e).copy()]
    # ties: S = 1.0 exactly, then values (k + 1/2) * 2^-52 for odd / even k, mixed with exact grid multiples
    t = [1.0] + [float(np.ldexp(2 * k + 1, -53)) for k in (0, 1, 2, 3, 1000, 1001, 4194303)] * 50 + [float(np.ldexp(1, -52))] * 7
    cases.append(np.array(t, dtype=np.float32))
    half = np.array([1.0] + [float(np.ldexp(1, -53))] * 5000, dtype=np.float32)     # every add is a tie: alternates stay / step
    cases.append(half)
    cases.append(np.array([0.5] * 3 + [float(np.ldexp(1, -54))] * 100 + [0.25, 0.25] + [float(np.ldexp(3, -54))] * 100, dtype=np.float32))
    cases.append(np.concatenate([np.zeros(5000, np.float32), np.array([1e-45, 1e-45, 3e-45, 1e-38, 1e-30], np.float32), e[:9000]]))
    cases.append(np.concatenate([e[:6000] * np.float32(1e-6), np.array([3.0e4], np.float32), e[:6000]]))
    cases.append(np.full(20000, 0.1, dtype=np.float32))
    cases.append(np.array([0.75, 0.25, 1.0, 2.0, 4.0, 8.0, 1e-9, 16.0 - 1e-6, 1e-6], dtype=np.float32))   # lands exactly on powers of two
    cases.append(rng.random(1, dtype=np.float32))
    cases.append(np.zeros(100, np.float32))
    cases.append((rng.random(70000, dtype=np.float32) * np.float32(2.0) ** rng.integers(-60, 3, 70000)).astype(np.float32))
    return cases
