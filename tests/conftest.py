import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (dict of numpy arrays, model kwargs, flavour)"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    rec = {k: z[k] for k in z.files}
    params = dict(ast.literal_eval(str(rec.pop("meta::params"))))
    flavour = "dis_embd3" if "dis_type" in params else "optim"
    return rec, params, flavour


def golden_state_dict(rec, torch):
    """Reference-named state_dict from a fixture (+ the shared STFT filters)."""
    filt = torch.from_numpy(np.load(os.path.join(GOLDEN, "stft_filters.npz"))["filters"])
    sd = {k[len("param::"):]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("param::")}
    sd["tfgridnet.enc.filterbank._filters"] = filt.clone()
    sd["tfgridnet.dec.filterbank._filters"] = filt.clone()
    return sd


def grad_fingerprint(np_grad, key, nproj=16):
    """as tests/golden/make_goldens.py: L2 norm and `nproj` seeded +-1 projections of a gradient tensor (float64)"""
    import zlib
    g = np.asarray(np_grad, np.float64).ravel()
    rng = np.random.default_rng(zlib.crc32(key.encode()))
    signs = rng.integers(0, 2, size=(nproj, g.size), dtype=np.int8) * 2 - 1
    return np.concatenate([[np.sqrt((g * g).sum())], signs @ g])


def build_default_ctor(factory, rec, torch):
    """A default_ctor_* fixture stores no weights: the model is rebuilt from the fixture's seed (same initialisers in the same
    order as the reference) and held to the stored per-tensor sums / norms.  factory: callable(**kwargs) -> module."""
    torch.manual_seed(int(rec["meta::seed"]))
    m = factory(L=4)
    sd = m.state_dict()
    for k, v in rec.items():
        if k.startswith("wsum::"):
            w = sd[k[len("wsum::"):]].double().numpy()
            np.testing.assert_allclose([w.sum(), np.sqrt((w * w).sum())], v, rtol=1e-9, atol=1e-9, err_msg=k)
    filt = torch.from_numpy(rec["filters"])
    m.load_state_dict({"tfgridnet.enc.filterbank._filters": filt.clone(), "tfgridnet.dec.filterbank._filters": filt.clone()},
                      strict=False)
    return m


def check_default_ctor_grads(named_grads, rec, tol):
    """full gradients where the fixture stores them (everything outside the blocks, first and last block), fingerprints for
    every tensor: the RMS over the projections of |delta| / |g| estimates the relative L2 error"""
    worst = ("", 0.0)
    for k, g in named_grads:
        g = np.asarray(g)
        fp_ref = rec["gfp::" + k]
        fp = grad_fingerprint(g, k)
        e = float(np.sqrt(((fp[1:] - fp_ref[1:]) ** 2).mean()) / (fp_ref[0] + 1e-30)) if fp_ref[0] > 0 else float(np.abs(g).max())
        if "grad::" + k in rec:
            ref = rec["grad::" + k]
            e = max(e, rel_l2(g, ref) if np.abs(ref).max() > 0 else float(np.abs(g).max()))
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < tol, worst
    return worst


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-30))


def flatten_state(d, prefix=""):
    out = {}
    for k in sorted(d.keys()):
        v = d[k]
        if isinstance(v, dict):
            out.update(flatten_state(v, prefix + k + "::"))
        else:
            out[prefix + k] = v.detach().cpu().numpy()
    return out


@pytest.fixture(scope="session")
def torch_mod():
    import torch
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    return torch


def poison_free_memory(torch, gib):
    """Every byte the caching allocator hands out next starts as NaN (one block of `gib` GiB filled and freed): a kernel that
    reads memory nobody wrote -- or that a not-yet-ordered launch was going to write -- shows up as NaN deterministically
    instead of depending on what the block held before (how the first-use bug of DESIGN.md 9.3 was pinned down)."""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t = torch.full((int(gib * (1 << 28)),), float("nan"), device="cuda")
    del t


@pytest.fixture(autouse=True)
def _poison_before_gpu_tests(request):
    """every -m gpu test starts on NaN-poisoned free memory (1 GiB block; the tests above that need more poison more)"""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        poison_free_memory(torch, 1)
    yield
