"""Host logic of the overlapped forward (no GPU): the order in which the next block's intra-frame tiles are taken and
the time slab of the inter-frame producer each of them waits for (sound_bubble_amd.ops.tile_order_np)."""
import numpy as np
import pytest


@pytest.mark.parametrize("B,T,slab", [(16, 625, 32), (2, 150, 32), (3, 129, 16), (1, 64, 32), (5, 17, 4)])
def test_tile_order_is_a_permutation_sorted_by_the_slab_that_completes_the_tile(B, T, slab):
    from sound_bubble_amd.ops import tile_order_np
    order, need = tile_order_np(B, T, slab)
    ntiles = (B * T + 15) // 16
    assert order.dtype == np.int32 and need.dtype == np.int32
    assert sorted(order.tolist()) == list(range(ntiles))
    assert np.all(np.diff(need) >= 0)                       # items are handed out in production order
    nslabs = (T + slab - 1) // slab
    assert need.min() >= 0 and need.max() == nslabs - 1
    for i, tile in enumerate(order):
        frames = [n for n in range(16 * tile, 16 * tile + 16) if n < B * T]       # frame n = (b, t) = divmod(n, T)
        slabs = [(n % T) // slab for n in frames]
        # every row of the tile's frames has been published once slab need[i] of ALL producer tiles is counted in
        assert need[i] == max(slabs)
    # a tile that straddles two batch entries (.., (b, T-1), (b+1, 0), ..) waits for the last slab
    if B > 1 and T % 16:
        straddle = [i for i, tile in enumerate(order) if (16 * tile) // T != min(16 * tile + 15, B * T - 1) // T]
        assert straddle and all(need[i] == nslabs - 1 for i in straddle)

