"""the build-owned counter-based noise stream (diffusion-ccsp_amd/noise.py)"""
import numpy as np
import pytest

from conftest import oracle, worlds
from diffusion_ccsp_amd import noise


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = noise.philox4x32_10(*ctr, *key)
        assert tuple(int(v) for v in got) == want


def test_stream_shapes_and_row_offsets():
    z = noise.normal(5, 3, 9, 5)
    assert z.shape == (9, 5) and z.dtype == np.float32
    # a shard that owns rows [4, 9) regenerates exactly its slice
    assert np.array_equal(noise.normal(5, 3, 5, 5, row0=4), z[4:])
    s = noise.normal_stream(5, 7, 9, 4)
    assert np.array_equal(s[3], noise.normal(5, 3, 9, 4))
    u = noise.uniform_stream(5, 50, 9)
    assert u.min() >= 0.0 and u.max() < 1.0
    big = noise.normal_stream(11, 400, 64, 4)
    assert abs(big.mean()) < 0.02 and abs(big.std() - 1.0) < 0.02


def test_call_counts():
    assert noise.n_normal_calls(1000, 10) == 11001
    assert noise.n_normal_calls(4, np.array([1, 2, 3, 4])) == 1 + 4 + 10


def test_oracle_c_stream_matches_numpy():
    """oracle/ccsp_oracle.c regenerates the same draws: a chain run with PHILOX(seed) equals the
    same chain run with the numpy stream INJECTED"""
    from conftest import oracle_model
    m = oracle_model('qualitative', 64, 'weights_qualitative_h64.npz', T=1000, S=2)
    b = worlds.qualitative_batch(1, 3, seed=2).to_torch()
    g = m.graph(b)
    N = b.x.shape[0]
    a = g.chain('ULA', seed=9, t_last=990)
    stream = noise.normal_stream(9, 1 + 10 * 3, N, 4)
    c = g.chain('ULA', normal=stream, t_last=990)
    assert np.array_equal(a, c)


@pytest.mark.gpu
def test_device_stream_matches_numpy(device):
    """PHILOX mode on the device reproduces the numpy stream to a few ulp (the initial state is
    0.5 * draw 0), and a whole chain driven by device draws equals the same chain with the numpy
    stream injected (contractive weights)"""
    import torch
    from conftest import weights
    from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion
    W = dict(weights('weights_qualitative_h64.npz'))
    den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=64, input_mode='qualitative', device=device)
    den.load_state_dict(W)
    gd = GaussianDiffusion(den, timesteps=1000, EBM='ULA', samples_per_step=3)
    b = worlds.qualitative_batch(7, 8, seed=3).to_torch()
    N = b.x.shape[0]
    x1, h1 = gd.p_sample_loop(b, return_history=True, seed=1234)
    stream = torch.from_numpy(noise.normal_stream(1234, gd.n_normal_calls(), N, 4))
    x2, h2 = gd.p_sample_loop(b, return_history=True, noise=stream)
    torch.cuda.synchronize()
    # initial state = 0.5 * z0 (free rows): compares the device draw itself
    z0 = noise.normal(1234, 0, N, 4)
    free = b.mask.numpy() == 0
    assert np.abs(h1[0].cpu().numpy()[free] - 0.5 * z0[free]).max() < 2e-6
    assert np.abs(x1.cpu().numpy() - x2.cpu().numpy()).max() < 1e-4
