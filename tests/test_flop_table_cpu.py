"""bench.py's `train_roofline.executed_*` comes from encoder.TrunkPlan.flop_table -- the per-layer, per-direction kernel choice the
trunk's own forward / backward schedule asks (TrunkPlan.kernel_for).  Here the table is held against an INDEPENDENT enumeration of
the encoder (models/actor_resnet.py:73-107: 3x3 stride-2 stem, four stages of two BasicBlocks, first block of every stage stride 2
with a 1x1 shortcut) priced by what each kernel family reports it executes (Winograd F(2x2,3x3): 16 multiplies per 2x2 output tile
and channel pair, the separate-pass GEMMs over the library's padded tile count), at the bench size (256 x 256) and the reference's
own training size (128 x 128, datasets/FiveKdataset.py:68).  (the hand-kept formula had gone stale.)"""
import pytest
import torch

from t2onet_amd import _lib
from t2onet_amd.actor_resnet import ResNet


def _expected(N, size):
    """[(layer, direction, family, algorithmic, executed)] written out from the shapes, not from the plan."""
    lib = _lib.load()
    rows = []
    h = size // 2
    f = 2.0 * 27 * 64 * N * h * h
    rows += [('stem', d, 'stem', f, f) for d in ('fwd', 'dgrad', 'wgrad')]
    cin = 64
    blk = 0
    for cout in (64, 128, 256, 512):
        for stride in (2, 1):
            ho = h // stride
            for name, ci, s, hi in (('conv1', cin, stride, h), ('conv2', cout, 1, ho)):
                algo = 2.0 * 9 * ci * cout * N * ho * ho
                for d in ('fwd', 'dgrad', 'wgrad'):
                    if s == 2:
                        fast = ho % 8 == 0 and hi % 2 == 0 if d != 'wgrad' else ho % 4 == 0 and hi % 2 == 0
                        fam, ex = ('direct' if fast else 'generic'), algo
                    else:
                        tiles = N * (hi // 2) * (hi // 2)
                        onchip = hi % 16 == 0 and cout <= 256            # t2o_wino_fused / t2o_wino_wgrad: maps in 16 x 16 blocks
                        if onchip:
                            fam, ex = ('wino_wgrad' if d == 'wgrad' else 'wino_fused'), 2.0 * 16 * tiles * ci * cout
                        elif cout >= 256 and hi % 2 == 0:                # separate passes: >= 256 channels, even maps
                            fam, ex = 'wino_sep', 2.0 * 16 * lib.t2o_wino_padded_tiles(N, hi, hi) * ci * cout
                        else:
                            fam, ex = 'direct', algo
                    rows.append(('block%d.%s' % (blk, name), d, fam, algo, ex))
            if stride == 2:
                f = 2.0 * cin * cout * N * ho * ho
                rows += [('block%d.shortcut' % blk, d, 'conv1x1', f, f) for d in ('fwd', 'dgrad', 'wgrad')]
            h, cin, blk = ho, cout, blk + 1
    return rows


@pytest.mark.parametrize('size', [256, 128])
def test_flop_table_equals_the_independent_enumeration(size):
    N = 64
    plan = ResNet().to(memory_format=torch.channels_last).trunk_plan()
    got = plan.flop_table(N, size, size)
    want = _expected(N, size)
    assert [(r[0], r[1], r[2]) for r in got] == [(r[0], r[1], r[2]) for r in want]
    for g, w in zip(got, want):
        assert g[3] == w[3] and g[4] == w[4], (g, w)
    algo = sum(r[3] for r in got)
    # 4.52 GFLOP per image forward at 256 x 256, x 3 directions (the fc layer and the stem's missing data gradient
    # into a 3-channel image are inside its 1 %)
    assert abs(algo / 3 / N - 4.52e9 * (size / 256.0) ** 2) < 0.01 * 4.52e9 * (size / 256.0) ** 2
    if size == 256:
        # 12 stride-1 layers, all three directions in the Winograd domain -> 2.41 TFLOP per bs = 64 episode step
        executed = 5 * sum(r[4] for r in got)
        assert abs(executed - 2.406e12) < 0.005e12
        assert sum(1 for r in got if r[2].startswith('wino')) == 36


def test_bench_accounting_reads_the_table():
    """bench.train_step_bench prices the step from flop_table (no second formula to go stale)."""
    import inspect
    import bench
    src = inspect.getsource(bench.train_step_bench)
    assert 'flop_table' in src and '16.0 / 36.0' not in src
