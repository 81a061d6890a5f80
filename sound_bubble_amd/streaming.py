"""Streaming causal inference (BASELINE config 5): 8 ms chunks, carried state, hipGraph-captured chunk step.

Mirrors edge/causal_infer.py: `ModelWrapper.feed` (:15-26) keeps `internal_state` and calls
`model(dict(mixture=frame), state, pad=False)`; `streaming_inference` (:28-47) rolls a
[1, M, chunk+pad] window by `chunk` samples per call.  Here the whole per-chunk launch sequence
(~10 kernels per block) is captured ONCE into a hipGraph; each `feed` is one copy of the 288-sample
window + one graph replay, and the recurrent/conv/iSTFT state lives in static device buffers that the
captured graph updates in place.
"""
import torch


def flatten_state(d, prefix=""):
    """'::'-joined names in sorted order (edge/flatbuf.py:8-25)."""
    out = {}
    for k in sorted(d):
        if isinstance(d[k], dict):
            out.update(flatten_state(d[k], prefix + k + "::"))
        else:
            out[prefix + k] = d[k]
    return out


DELIMITER = "::"


def flatten_state_buffers(state_buffers_dict, prefix=""):
    """edge/flatbuf.py:10-25: (names, buffers) of a nested state dict, keys sorted at every level, names joined with
    '::', every buffer cloned -- the chunk-level I/O order of the exported edge model (to_onnx.py:94-101)."""
    flat = flatten_state(state_buffers_dict, prefix)
    for k, v in flat.items():
        if not torch.is_tensor(v):
            raise TypeError(f"{k}: expected torch.Tensor, found {type(v).__name__}")
    return list(flat.keys()), [v.clone() for v in flat.values()]


def unflatten_state_buffers(state_names, state_buffers):
    """edge/flatbuf.py:27-70: rebuild the nested state dict from '::'-joined names (buffers cloned)."""
    if len(state_names) != len(state_buffers):
        raise ValueError(f"{len(state_names)} names for {len(state_buffers)} buffers")
    root = {}
    for name, buf in zip(state_names, state_buffers):
        path = name.split(DELIMITER)
        node = root
        for part in path[:-1]:
            nxt = node.setdefault(part, {})
            if not isinstance(nxt, dict):
                raise ValueError(f"{name}: '{part}' is both a buffer and a group")
            node = nxt
        if isinstance(node.get(path[-1]), dict):
            raise ValueError(f"{name}: both a buffer and a group")
        node[path[-1]] = buf.clone()
    return root


def _clone_tree(d):
    return {k: _clone_tree(v) if isinstance(v, dict) else v for k, v in d.items()}


# Parameters REPLACED on any module (`module.weight = nn.Parameter(...)`, `load_state_dict(assign=True)` on a sub-module) go
# through nn.Module.register_parameter: a global registration hook counts them, so that a graphed separator notices on its
# very next feed (one integer compare per chunk) that the addresses its graph was captured with may be gone.
_PARAM_REGISTRATIONS = [0]


def _count_registration(module, name, param):
    _PARAM_REGISTRATIONS[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_count_registration)


class StreamingSeparator(torch.nn.Module):
    """A captured hipGraph replays against fixed ADDRESSES: the parameters, the state buffers and the per-model inference
    workspaces of the capture.  The separator therefore (i) pins the workspaces its capture used (they are never evicted or
    re-zeroed while the graph lives: functional.Workspaces), (ii) re-captures when a parameter object was replaced -- noticed
    on the next feed through the registration counter above and the model's load_state_dict hook -- and (iii) offers
    `invalidate()` for the one case nothing can observe: re-pointing `p.data` of an existing parameter (FlatBucket does that;
    build the separator afterwards, or call invalidate())."""

    def __init__(self, model, batch_size=1, dis_embed=None, use_graph=True):
        super().__init__()
        import weakref
        self.model = model.eval()
        dev = next(model.parameters()).device
        self.chunk, self.pad = model.stft_chunk_size, model.stft_pad_size
        self.frame = torch.zeros(batch_size, model.num_ch, self.chunk + self.pad, device=dev)
        self.state = model.init_buffers(batch_size, dev)          # static buffers, updated in place
        self.dis_embed = dis_embed.to(dev) if dis_embed is not None else None
        self.out = None
        self.graph = None
        self._param_key = None
        self._dirty = False
        self._seen_registrations = -1
        ps = list(model.parameters())
        self._ends, self._end_ptrs, self._feeds = (ps[0], ps[-1]), None, 0   # (Parameter objects: replaced ones are caught by the registration count)
        self._pinned = None                                       # (Workspaces, keys) held by the captured graph
        self.use_graph = use_graph
        me = weakref.ref(self)                                    # the hook must not keep the separator (and its graph) alive

        def _mark(*_):
            s = me()
            if s is not None:
                s._dirty = True
        self._hook = model.register_load_state_dict_post_hook(_mark)

    def invalidate(self):
        """the next feed re-captures the graph (call after re-pointing `p.data` of a parameter of the model)"""
        self._dirty = True
        self._param_key = None

    def _release(self):
        if self._pinned is not None:
            ws, keys = self._pinned
            ws.unpin(keys)
            self._pinned = None
        self.graph = None

    def __del__(self):
        try:
            self._release()
            self._hook.remove()
        except Exception:
            pass

    def _inputs(self):
        d = {"mixture": self.frame}
        if self.dis_embed is not None:
            d["dis_embed"] = self.dis_embed
        return d

    def _step_inplace(self):
        """one chunk; the new state is written back INTO the static state buffers"""
        static = flatten_state(self.state)
        st = _clone_tree(self.state)                               # shallow: same tensors, fresh dicts
        out = self.model(self._inputs(), st, pad=False)["output"]
        # `self.internal_state = next_state` (edge/causal_infer.py:24) into the static buffers: one launch for all of them
        pairs, odd = [], []
        for k, v in flatten_state(st).items():
            if v.data_ptr() != static[k].data_ptr():
                ok = v.is_contiguous() and v.dtype == torch.float32 and v.numel() == static[k].numel()
                (pairs if ok else odd).append((v, static[k]))
        if pairs:
            from . import ops
            ops.multi_copy(pairs)
        for v, dst in odd:
            dst.copy_(v)
        return out

    def _capture(self):
        from . import functional as Fn
        self._release()
        static = flatten_state(self.state)
        snap = {k: v.clone() for k, v in static.items()}
        ws = self.model.__dict__.setdefault("_ws", Fn.Workspaces())
        ws.record()                                                # every workspace the chunk step touches from here on ...
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():                # warm-up off-graph (allocator, lazy init)
            for _ in range(2):
                self._step_inplace()
        torch.cuda.current_stream().wait_stream(s)
        for k, v in static.items():                                # undo the warm-up's state updates
            v.copy_(snap[k])
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = self._step_inplace()
        self._pinned = (ws, ws.pin_recorded())                     # ... stays allocated for as long as this graph lives
        for k, v in static.items():
            v.copy_(snap[k])

    @torch.no_grad()
    def feed(self, frame):
        """frame: [B, M, chunk+pad] window (already rolled).  Returns [B, 1, chunk] (a static buffer when graphed)."""
        self.frame.copy_(frame, non_blocking=True)
        if not self.use_graph:
            return self._step_inplace()
        # the graph holds the addresses of the parameters: re-capture when one was replaced (load_state_dict(assign=True),
        # module.weight = ...); in-place updates need nothing -- the weight-form refresh is one of the captured launches.
        # Walking all parameters costs ~50 us of Python, a sixth of a chunk: done only when a parameter was registered
        # somewhere since the last look (one integer compare per chunk) or a load_state_dict ran on the model.
        # ... and, cheaply, every chunk: the addresses of the first and the last parameter (model.to() / .half() / a FlatBucket built
        # after this separator re-point p.data through neither hook; all parameters move together in those cases), with the full
        # walk every 256th chunk as the backstop (a replay against freed weights is silently wrong audio)
        self._feeds += 1
        ends = (self._ends[0].data_ptr(), self._ends[1].data_ptr())
        if (self.graph is None or self._dirty or self._seen_registrations != _PARAM_REGISTRATIONS[0] or ends != self._end_ptrs
                or self._feeds % 256 == 0):
            self._end_ptrs = ends
            self._seen_registrations = _PARAM_REGISTRATIONS[0]
            key = tuple((id(p), p.data_ptr()) for p in self.model.parameters())
            if self.graph is None or key != self._param_key:
                self._capture()
                self._param_key = key
            self._dirty = False
        self.graph.replay()
        return self.out

    def reset(self):
        for v in flatten_state(self.state).values():
            v.zero_()


@torch.no_grad()
def streaming_inference(sep, X):
    """edge/causal_infer.py:28-47: X [B, M, k*chunk + pad] -> [B, 1, k*chunk]."""
    T, P = sep.chunk, sep.pad
    cur = torch.zeros(X.shape[0], X.shape[1], T + P, device=X.device)
    cur[..., -P:] = X[..., :P]
    outs = []
    for i in range(P, X.shape[-1] - P + 1, T):
        cur = torch.roll(cur, shifts=-T, dims=-1)
        cur[..., -T:] = X[..., i:i + T]
        outs.append(sep.feed(cur).clone())
    return torch.cat(outs, dim=-1)
