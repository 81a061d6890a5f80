"""The dispatcher-registered operators (sound_bubble_amd/torch_ops.py) on the CPU box: schemas, fake (meta) implementations
under FakeTensorMode with `cuda` fake tensors -- shape propagation through separator, losses and the backward operator needs
no GPU and no kernel -- and that the real implementations are registered for the GPU only (no CPU fallback)."""
import pytest

from conftest import load_golden


@pytest.mark.parametrize("name,cls", [("tiny_big", "NetDisEmbd3"), ("tiny_small", "NetOptim"),
                                      ("tiny_big_attn100", "NetDisEmbd3")])
def test_fake_implementations_propagate_shapes(torch_mod, name, cls):
    torch = torch_mod
    import sound_bubble_amd as sb
    from sound_bubble_amd import torch_ops as T
    from sound_bubble_amd.streaming import flatten_state
    from torch._subclasses.fake_tensor import FakeTensorMode
    rec, params, flavour = load_golden(name)
    m = getattr(sb, cls)(**params)
    mid = T.register_model(m)
    assert T.register_model(m) == mid
    B, N = 3, 1000
    want_state = {k: tuple(v.shape) for k, v in flatten_state(m.init_buffers(B, "cpu")).items()}
    with FakeTensorMode():
        mix = torch.empty(B, 6, N, device="cuda")
        dis = torch.empty(B, 3, device="cuda") if flavour == "dis_embd3" else None
        ps = [torch.empty(p.shape, device="cuda") for p in m.parameters()]
        outs = torch.ops.sound_bubble.separate(mix, dis, ps, [], mid, True, False)
        assert tuple(outs[0].shape) == (B, 1, N) and outs[0].device.type == "cuda" and outs[0].dtype == torch.float32
        assert [tuple(o.shape) for o in outs[1:-1]] == list(want_state.values())
        assert outs[-1].device.type == "cpu" and outs[-1].dtype == torch.int64
        # streaming call: pad=False, carried state in, hop-multiple out
        look, hop = m.stft_pad_size, m.stft_chunk_size
        o2 = torch.ops.sound_bubble.separate(torch.empty(B, 6, look + 3 * hop, device="cuda"), dis, ps, list(outs[1:-1]), mid,
                                             False, False)
        assert tuple(o2[0].shape) == (B, 1, 3 * hop)
        with pytest.raises(RuntimeError):
            torch.ops.sound_bubble.separate(torch.empty(B, 6, look + hop + 1, device="cuda"), dis, ps, [], mid, False, False)
        loss, lv, d = torch.ops.sound_bubble.snrlp_loss(outs[0], torch.empty_like(outs[0]), 100.0)
        assert loss.shape == () and tuple(lv.shape) == (B,) and d.shape == outs[0].shape
        l2, d2 = torch.ops.sound_bubble.multireso_fuse_loss(outs[0], torch.empty_like(outs[0]), "{}")
        assert l2.shape == () and d2.shape == outs[0].shape
        gs = torch.ops.sound_bubble.separate_backward(outs[-1], outs[0], mid)
        assert [tuple(g.shape) for g in gs] == [tuple(p.shape) for p in m.parameters()]


def test_schemas_and_gpu_only_registration(torch_mod):
    torch = torch_mod
    import sound_bubble_amd as sb
    from sound_bubble_amd import torch_ops as T
    s = str(torch.ops.sound_bubble.separate.default._schema)
    assert s.startswith("sound_bubble::separate(Tensor mixture, Tensor? dis_embed, Tensor[] params, Tensor[] state,") \
        and s.endswith("-> Tensor[]"), s
    assert "-> (Tensor, Tensor, Tensor)" in str(torch.ops.sound_bubble.snrlp_loss.default._schema)
    assert "-> (Tensor, Tensor)" in str(torch.ops.sound_bubble.multireso_fuse_loss.default._schema)
    # no CPU implementation: the product fails loudly off the GPU
    x = torch.zeros(1, 1, 192)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.sound_bubble.snrlp_loss(x, x, 1.0)
    rec, params, _ = load_golden("tiny_small")
    m = sb.NetOptim(**params)
    with pytest.raises((NotImplementedError, RuntimeError)):
        T.separate_module(m)({"mixture": torch.zeros(1, 6, 960)})
    # the backward operator declares the flat gradient bucket it adds into
    sb_ = str(torch.ops.sound_bubble.separate_backward_bucket.default._schema)
    assert "!) grad_bucket" in sb_ and sb_.endswith("-> Tensor"), sb_
    from torch._higher_order_ops.auto_functionalize import can_auto_functionalize
    assert can_auto_functionalize(torch.ops.sound_bubble.separate_backward_bucket.default)     # a compiler can take the mutation
    # unknown model id
    with pytest.raises(RuntimeError):
        T._model(10 ** 9)
