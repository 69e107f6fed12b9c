"""TouchDataset reader + low-level datapipe: the oracle and the product against what the REFERENCE's
reader / datapipe returned (tests/golden/touchdataset.npz) on shards written with the reference's own IndexWriter
from its test assets (tests/golden/touchdataset/, md5-identical to the reference test's constants)."""
import hashlib
import os
import types

import numpy as np
import pytest
import torch

from oracle import dataset as ods
from touchnet_amd.data.datapipe import LowLevelTouchDatapipe, MidLevelTouchDatapipe
from touchnet_amd.data.dataset import TouchDataset

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "touchdataset")
ONE = [os.path.join(ROOT, "1sample_per_shard", f"00000000{i}") for i in (0, 1)]
TWO = [os.path.join(ROOT, "2sample_per_shard", "000000000")]
SYN = [os.path.join(ROOT, "synthetic", f"00000000{i}") for i in (0, 1)]
CASES = {   # name -> (list dirs, dp_rank, dp_world, start state, config overrides)   == make_golden.py::touchdataset_case
    "plain_1per": (ONE, 0, 1, None, {}),
    "plain_2per": (TWO, 0, 1, None, {}),
    "shuffled_2epochs": (SYN + ONE, 0, 1, None, dict(datalist_epoch=2, datalist_shuffling=True, dataset_shuffling=True)),
    "sharded_rank1of2": (SYN + ONE, 1, 2, None, dict(datalist_sharding=True, datalist_shuffling=True)),
    "segments": (SYN, 0, 1, None, dict(dataset_load_audio_via_segments=True, dataset_shuffling=True)),
    "random_cut": (SYN + TWO, 0, 1, None, dict(dataset_random_cut_audio=True, dataset_random_cut_audio_min_length_in_ms=500,
                                               dataset_random_cut_audio_max_length_in_ms=1500)),
    "resumed": (SYN, 0, 1, dict(epoch=0, consumed_lists=0, consumed_samples=2), dict(dataset_shuffling=True)),
}


def _cfg(tmp_path, dirs, **over):
    lst = tmp_path / "data.list"
    lst.write_text("".join(f"{d} audio+metainfo\n" for d in dirs))
    cfg = types.SimpleNamespace(datalist_path=str(lst), datalist_epoch=1, datalist_shuffling=False,
                                datalist_sharding=False, dataset_mmap=True, dataset_shuffling=False,
                                dataset_load_audio_via_segments=False, dataset_random_cut_audio=False,
                                dataset_random_cut_audio_min_length_in_ms=5000,
                                dataset_random_cut_audio_max_length_in_ms=3600000)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def test_fixture_md5_equals_reference_test_constants():
    """tests/touchnet/bin/test_make_data.py:25-28,42-45: md5 over the sorted md5s of every .idx/.bin of a layout."""
    expect = {"1sample_per_shard": "05fe272d67459992748bbf5720c5a92e", "2sample_per_shard": "93245372eca0dce2013c1e5bd393f17f"}
    for layout, want in expect.items():
        sums = []
        for dirpath, _, files in os.walk(os.path.join(ROOT, layout)):
            for f in files:
                if f.endswith((".idx", ".bin")):
                    p = os.path.join(dirpath, f)
                    sums.append((hashlib.md5(open(p, "rb").read()).hexdigest(), p))
        # `md5sum | sort | cut -f1` sorts the "<hash>  <path>" lines
        lines = sorted(f"{h}  {p}" for h, p in sums)
        got = hashlib.md5("".join(ln.split(" ")[0] + "\n" for ln in lines).encode()).hexdigest()
        assert got == want, layout


def _layout_md5(root):
    sums = []
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".idx", ".bin")):
                p = os.path.join(dirpath, f)
                sums.append((hashlib.md5(open(p, "rb").read()).hexdigest(), p))
    lines = sorted(f"{h}  {p}" for h, p in sums)
    return hashlib.md5("".join(ln.split(" ")[0] + "\n" for ln in lines).encode()).hexdigest()


def test_writer_reproduces_the_reference_tests_md5_constants(tmp_path):
    """touchnet_amd/data/builder.py (the role of make_data.py's DataBuilder + IndexWriter): the two reference test
    utterances — read back from the fixture shards with the product reader — written again with the product writer, one
    and two per shard, give the md5s the reference's own test pins (tests/touchnet/bin/test_make_data.py:25-28), and every
    file equals the fixture's byte for byte."""
    import json
    from touchnet_amd.data.builder import DataBuilder, write_audio_shards
    from touchnet_amd.data.dataset import TouchDataset
    samples = []
    for shard in ("000000000", "000000001"):
        ds = TouchDataset(os.path.join(ROOT, "1sample_per_shard", shard), True, "audio+metainfo")
        meta = json.loads(ds.get(0, "metainfo").tobytes().decode("utf-8"))
        assert meta.pop("sample_rate") == 16000
        samples.append((meta, np.array(ds.get(0, "audio"))))
    expect = {1: "05fe272d67459992748bbf5720c5a92e", 2: "93245372eca0dce2013c1e5bd393f17f"}
    for per, want in expect.items():
        out = str(tmp_path / f"{per}sample_per_shard")
        shards = write_audio_shards(samples, out, per)
        assert _layout_md5(out) == want
        for d in shards:
            for f in os.listdir(d):
                ref = os.path.join(ROOT, f"{per}sample_per_shard", os.path.basename(d), f)
                assert open(os.path.join(d, f), "rb").read() == open(ref, "rb").read(), (per, d, f)
    # add_document: several sequences of one document in one call, read back by the product reader
    b = DataBuilder(str(tmp_path / "texttoken.bin"), np.uint16)
    b.add_document(np.arange(7), [3, 4])
    b.add_item(np.array([9, 8]))
    b.end_document()
    b.finalize(str(tmp_path / "texttoken.idx"))
    ds = TouchDataset(str(tmp_path), True, "texttoken")
    assert len(ds) == 3 and ds.get(1, "texttoken").tolist() == [3, 4, 5, 6] and ds.get(2, "texttoken").tolist() == [9, 8]
    assert ds.index["texttoken"].document_indices.tolist() == [0, 2, 3]


def test_reader_random_access(golden):
    g = golden("touchdataset.npz")
    ds = TouchDataset(TWO[0], True, "audio+metainfo")
    assert len(ds) == int(g["reader/len"])
    for i in (0, 1):
        assert [int(v) for v in ds.get_idx(i, "audio")] == g[f"reader/idx{i}"].tolist()
    part = ds.get(1, "audio", offset=1000, length=64)
    assert part.dtype == np.int16
    np.testing.assert_array_equal(part, g["reader/partial"])
    np.testing.assert_array_equal(ds.get(0, "metainfo"), g["reader/meta0"])
    # oracle restatement agrees
    np.testing.assert_array_equal(ods.read_item(TWO[0], "audio", 1, 1000, 64), g["reader/partial"])
    np.testing.assert_array_equal(ods.read_item(TWO[0], "metainfo", 0), g["reader/meta0"])
    with pytest.raises(FileNotFoundError):
        TouchDataset(os.path.join(ROOT, "nope"))


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("pcm16", [False, True])
def test_datapipe_order_and_samples(golden, tmp_path, case, pcm16):
    g = golden("touchdataset.npz")
    dirs, rank, world, state, over = CASES[case]
    cfg = _cfg(tmp_path, dirs, dataset_keep_pcm16=pcm16, **over)
    pipe = LowLevelTouchDatapipe(cfg, rank, world)
    if state:
        pipe.load_state_dict(state)
    keys, txts, lens, sums, heads, states = [], [], [], [], [], []
    for smp in pipe:
        w = smp["waveform"]
        assert w.dim() == 2 and w.shape[0] == 1 and smp["datatypes"] == "audio+metainfo"
        if pcm16:
            assert w.dtype == torch.int16
            pcm = w[0].numpy().astype(np.int64)
        else:
            assert w.dtype == torch.float32
            pcm = (w[0].numpy() * 32768.0).astype(np.int64)
        keys.append(smp["key"]); txts.append(smp["txt"]); lens.append(pcm.size)
        sums.append(int(np.abs(pcm).sum())); heads.append(pcm[:4].tolist())
        st = pipe.state_dict()
        states.append([st["epoch"], st["consumed_lists"], st["consumed_samples"]])
    assert keys == g[f"{case}/keys"].tolist() and txts == g[f"{case}/txts"].tolist()
    assert lens == g[f"{case}/lens"].tolist() and sums == g[f"{case}/abs_sums"].tolist()
    assert heads == g[f"{case}/heads"].tolist() and states == g[f"{case}/states"].tolist()


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_datapipe_pinned_by_reference(golden, case):
    g = golden("touchdataset.npz")
    dirs, rank, world, state, over = CASES[case]
    cfg = types.SimpleNamespace(datalist_epoch=1, datalist_shuffling=False, datalist_sharding=False,
                                dataset_shuffling=False, dataset_load_audio_via_segments=False,
                                dataset_random_cut_audio=False, dataset_random_cut_audio_min_length_in_ms=5000,
                                dataset_random_cut_audio_max_length_in_ms=3600000)
    for k, v in over.items():
        setattr(cfg, k, v)
    st = (state["epoch"], state["consumed_lists"], state["consumed_samples"]) if state else (0, 0, 0)
    got = list(ods.iterate([(d, "audio+metainfo") for d in dirs], cfg, rank, world, st))
    assert [m["key"] for _, m in got] == g[f"{case}/keys"].tolist()
    assert [m["txt"] for _, m in got] == g[f"{case}/txts"].tolist()
    assert [m["pcm"].size for _, m in got] == g[f"{case}/lens"].tolist()
    assert [int(np.abs(m["pcm"].astype(np.int64)).sum()) for _, m in got] == g[f"{case}/abs_sums"].tolist()
    assert [list(s) for s, _ in got] == g[f"{case}/states"].tolist()


def test_midlevel_composes(tmp_path):
    cfg = _cfg(tmp_path, TWO)

    def tag(it, label):
        for s in it:
            s["tag"] = label
            yield s
    pipe = MidLevelTouchDatapipe(LowLevelTouchDatapipe(cfg, 0, 1), tag, "x")
    out = list(pipe)
    assert len(out) == 2 and all(s["tag"] == "x" for s in out)
    assert pipe.state_dict() == {"epoch": 1, "consumed_lists": 0, "consumed_samples": 0}
