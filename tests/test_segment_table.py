"""The optimizer's segment / chunk tables (ops/optim.py:SegmentTable), the host side of csrc/optim.h: row layout of a
SegEntry (10 ints, the last two = the transposed shadow copy), one chunk per weight row where the row crosses LDS,
64-element chunks for slab-summed segments, 2048-element chunks otherwise, and subset()."""
import torch

from pytorch_distributed_amd.ops.optim import OPT_CHUNK, OPT_ROW_LDS, SegmentTable, plain_table


def _entries():
    return [
        {"start": 0, "count": 64 * 512, "cols": 512, "C": 32, "P": 16, "shadow_off": 0, "shadow_ld": 512, "shadow": True,
         "fine": True, "tr_off": 100000, "tr_ld": 64},                        # conv2-like, slab-summed, transposed copy
        {"start": 64 * 512, "count": 64, "fine": True},                        # its bias
        {"start": 64 * 512 + 64, "count": 8 * 3136, "cols": 3136, "C": 64, "P": 49, "shadow_off": 64 * 512,
         "shadow_ld": 3136, "shadow": True},                                   # fc1-like rows
        {"start": 64 * 512 + 64 + 8 * 3136, "count": 5000},                    # plain fp32 tensor
    ]


def test_rows_mirror_the_device_struct():
    t = SegmentTable(_entries(), torch.device("cpu"))
    assert t.segs.dtype == torch.int32 and tuple(t.segs.shape) == (4, 10)      # struct SegEntry: 10 ints
    r = t.segs[0].tolist()
    assert r == [0, 64 * 512, 512, 32, 16, 0, 512, 1 | 2, 100000, 64]
    assert t.segs[1].tolist()[7] == 2 and t.segs[1].tolist()[8:] == [0, 0]
    assert t.segs[2].tolist()[7] == 1 and t.segs[3].tolist()[2:5] == [5000, 1, 5000]


def test_chunking_rules():
    t = SegmentTable(_entries(), torch.device("cpu"))
    by_seg = {}
    for seg, first in t.chunks.tolist():
        by_seg.setdefault(seg, []).append(first)
    assert (16 | 1) * 32 <= OPT_ROW_LDS and (49 | 1) * 64 <= OPT_ROW_LDS
    assert by_seg[0] == list(range(0, 64 * 512, 512))          # a shadowed C > 1 segment: one row per workgroup
    assert by_seg[1] == [0]                                    # fine: 64-element chunks
    assert by_seg[2] == list(range(0, 8 * 3136, 3136))
    assert by_seg[3] == list(range(0, 5000, OPT_CHUNK))
    assert t.nchunks == len(t.chunks) == 64 + 1 + 8 + 3


def test_subset_keeps_rows_and_filters_chunks():
    t = SegmentTable(_entries(), torch.device("cpu"))
    tail = t.subset(lambda s: s >= 2)
    assert tail.segs is t.segs
    assert tail.nchunks == 8 + 3 and all(seg >= 2 for seg, _ in tail.chunks.tolist())
    none = t.subset(lambda s: False)
    assert none.nchunks == 0 and tuple(none.chunks.shape) == (1, 2)           # a placeholder row, never launched


def test_plain_table():
    t = plain_table(5000, torch.device("cpu"))
    assert t.segs.tolist() == [[0, 5000, 5000, 1, 5000, 0, 0, 0, 0, 0]]
    assert t.chunks.tolist() == [[0, 0], [0, OPT_CHUNK], [0, 2 * OPT_CHUNK]]
