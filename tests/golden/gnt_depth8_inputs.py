"""Seeded inputs and weights of the depth-8 GNT fixture (gnt_depth8.npz): pure numpy, imported by the generator
(make_golden_gnt_depth8.py, which feeds them to the REFERENCE's GNT) and by the tests (which feed them to the oracle, the
torch mirror and the HIP kernels).  Only the reference's OUTPUTS are stored in the fixture; the 3.5 MB of weights and the
inputs are regenerated here, and the fixture carries their checksums so that a generator that drifts fails loudly.

The network is the one the reference runs (configs/static_renderer/gnt.yaml:9 transformer_depth 8, netwidth 64;
configs/engine/evaluator_pgdvs.yaml:14-16: 256 samples per ray; configs/_basic.yaml:45: 10 source views), parameters in the
order of ``GNT.state_dict()`` (pgdvs/models/gnt/models/transformer_network.py:341-421)."""
import numpy as np

CASES = {"v10": dict(R=16, Ss=256, V=10, seed=810), "v24": dict(R=8, Ss=256, V=24, seed=824)}
WEIGHT_SEED = 88


def make_weights(shapes):
    """shapes: ordered {name: shape} of the network's state_dict.  Linear weights / biases uniform in +-1/sqrt(fan_in) (torch's
    default scale), LayerNorm weights 1 + 0.1 n, LayerNorm / 1-d biases get an extra 0.1 n so that mistakes show."""
    rng = np.random.default_rng(WEIGHT_SEED)
    out = {}
    fan_in = 1
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        if "norm" in name and name.endswith("weight"):
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif "norm" in name:
            w = 0.1 * rng.standard_normal(shape)
        elif len(shape) == 2:
            fan_in = shape[1]
            w = rng.uniform(-1.0, 1.0, shape) / np.sqrt(fan_in)
            if "view_selftrans" in name and (".attn.q_fc" in name or ".attn.k_fc" in name):
                w *= 3.0  # peaked ray attention (random-init scores are ~uniform over the 256 samples otherwise)
        else:  # bias of the Linear just before it in state_dict order
            w = rng.uniform(-1.0, 1.0, shape) / np.sqrt(fan_in) + 0.1 * rng.standard_normal(shape)
        out[name] = w.astype(np.float32)
    return out


def make_inputs(case):
    c = CASES[case]
    R, Ss, V = c["R"], c["Ss"], c["V"]
    rng = np.random.default_rng(c["seed"])
    # rays of a camera looking down +z, inverse-depth samples between 0.8 and 6 (ray_sampler.py:59-73)
    ray_o = rng.normal(0, 0.05, (R, 3))
    ray_d = np.concatenate([rng.uniform(-0.6, 0.6, (R, 2)), np.ones((R, 1))], 1)
    z = 1.0 / np.linspace(1.0 / 0.8, 1.0 / 6.0, Ss)
    pts = ray_o[:, None] + ray_d[:, None] * z[None, :, None]
    # colours in [0, 1], ResUNet-like features (a shared per-sample part + a per-view part: views agree more or less)
    rgb = np.clip(rng.uniform(0, 1, (R, Ss, 1, 3)) + rng.normal(0, 0.15, (R, Ss, V, 3)), 0, 1)
    feat = rng.normal(0, 0.7, (R, Ss, 1, 32)) + rng.normal(0, 0.5, (R, Ss, V, 32))
    rgb_feat = np.concatenate([rgb, feat], -1)
    # ray_diff: unit difference of directions + their dot product (gnt/projector.py:75-115)
    d = rng.normal(0, 1, (R, Ss, V, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    dot = rng.uniform(0.6, 1.0, (R, Ss, V, 1))
    ray_diff = np.concatenate([d, dot], -1)
    # masks: ~70 % valid; ray 0 sees no view at all, ray 1 exactly one (a different one per sample), ray 2 all;
    # rays 3.. have stretches of samples without any view (out of every frustum near the camera)
    mask = (rng.uniform(0, 1, (R, Ss, V, 1)) < 0.7)
    mask[0] = False
    mask[1] = False
    mask[1, np.arange(Ss), np.arange(Ss) % V] = True
    mask[2] = True
    for r in range(3, R):
        mask[r, : int(rng.integers(0, Ss // 4))] = False
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(rgb_feat=f32(rgb_feat), ray_diff=f32(ray_diff), mask=f32(mask), pts=f32(pts), ray_d=f32(ray_d))


def checksum(arrs):
    """order-sensitive float64 digest of a dict of arrays"""
    s = 0.0
    for i, (k, a) in enumerate(arrs.items()):
        a = np.asarray(a, np.float64).ravel()
        s += float((a * np.cos(np.arange(a.size) * 0.37 + i)).sum())
    return s
