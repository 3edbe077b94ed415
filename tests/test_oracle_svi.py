"""CPU: the oracle's restatement of the in-place SVI draw's generator (oracle/bnn_oracle.py: philox4x32_10, philox_normals,
svi_draw_philox) is pinned by the known-answer vectors Random123 publishes for Philox4x32-10 (kat_vectors: counter, key -> output),
and its Box-Muller layer by its moments.  The HIP kernel is compared with this restatement element by element in
tests/test_hip_svi.py.  The draw as a whole stays PARITY UNPINNED against pyro-ppl 1.3.0 (absent)."""
import numpy as np
import torch

from oracle import bnn_oracle as O

KAT = [  # Random123 kat_vectors, "philox4x32 10": c0 c1 c2 c3 k0 k1 -> out
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answers():
    for inp, want in KAT:
        got = O.philox4x32_10(*[np.uint32(v) for v in inp])
        assert tuple(int(v) for v in got) == want


def test_normals_layout_and_moments():
    n = O.philox_normals(key=0xABCDEF0123456789, draw_id=3, tensor_id=0, sample=2, rows=64, cols=784)
    assert n.shape == (64, 784) and abs(n.mean()) < 0.02 and abs(n.std() - 1) < 0.02 and np.abs(n).max() < 6.77
    # element (r, c) = component c % 4 of block (r * ceil(cols/4) + c // 4): the same eps whatever the row count / the padding
    m = O.philox_normals(key=0xABCDEF0123456789, draw_id=3, tensor_id=0, sample=2, rows=8, cols=784)
    assert np.array_equal(m, n[:8])
    odd = O.philox_normals(key=1, draw_id=0, tensor_id=4, sample=0, rows=3, cols=10)          # cols % 4 != 0: 3 blocks per row
    assert odd.shape == (3, 10) and not np.array_equal(odd[0], odd[1])
    # the sample index enters the counter unless the key is the sample's own seed
    a = O.philox_normals(7, 0, 0, 1, 4, 16)
    assert not np.array_equal(a, O.philox_normals(7, 0, 0, 2, 4, 16))
    assert np.array_equal(O.philox_normals(7, 0, 0, 1, 4, 16, sample_is_key=True), O.philox_normals(7, 0, 0, 5, 4, 16, sample_is_key=True))


def test_draw_is_loc_plus_softplus_scale_times_eps():
    loc = {"W1": torch.randn(8, 20), "b1": torch.randn(8), "W2": torch.randn(3, 8), "b2": torch.randn(3)}
    scl = {k: torch.randn_like(v) for k, v in loc.items()}
    W, E = O.svi_draw_philox(loc, scl, key=5, draw_id=1, n_samples=4)
    for k in loc:
        assert W[k].shape == (4,) + tuple(loc[k].shape)
        want = O.svi_materialize({k: loc[k].double()}, {k: scl[k].double()}, {k: E[k]})[k]        # model_bnn.py:124-130's rsample
        assert float((W[k] - want).abs().max()) < 1e-12
