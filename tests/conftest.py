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
