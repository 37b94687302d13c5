"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header
declares; the engine's launch sequence (shapes, strides, buffer bookkeeping) runs end to end
against a stub library that computes NOTHING (every launch returns 0); the device-side segment
map matches the oracle's restatement of the reference token loop."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import medmoe_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from medmoe_amd import lib_path
    if not os.path.exists(lib_path()):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(lib_path())
    hdr = open(os.path.join(ROOT, "include", "medmoe_hip.h")).read()
    names = re.findall(r"^int (medmoe_\w+)\(", hdr, re.M)
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n


def test_product_path_does_not_import_oracle():
    for f in os.listdir(os.path.join(ROOT, "medmoe_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "medmoe_amd", f)).read()
            assert not re.search(r"^\s*(import|from)\s+\S*oracle", src, re.M), f
            assert "medmoe_oracle" not in src and "_ref_import" not in src, f


class _StubLib:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        if not name.startswith("medmoe_"):
            raise AttributeError(name)

        def f(*a):
            self.calls.append(name)
            if name == "medmoe_local_geometry":
                HW, T = a[0].value, a[1].value
                a[2]._obj.value = (HW + 15) // 16 * 16; a[3]._obj.value = (T + 15) // 16 * 16
                a[4]._obj.value = (((HW + 15) // 16) + 1) // 2 * 32
            if name == "medmoe_local_fast_path":
                nht, ntt = (a[0].value + 15) // 16, (a[1].value + 15) // 16
                return int((nht == 4 and ntt == 1) or (nht in (13, 16) and 1 <= ntt <= 5))
            if name == "medmoe_local_pair3_supported":
                HW, ntt = a[0].value, (a[1].value + 15) // 16
                return int((HW == 64 and ntt == 1) or (HW == 196 and 1 <= ntt <= 5))
            return 0
        return f


@pytest.fixture
def stub(monkeypatch):
    from medmoe_amd import _lib, ops
    lib = _StubLib()
    monkeypatch.setattr(_lib, "_LIB", lib)
    monkeypatch.setattr(ops, "load_library", lambda: lib)
    monkeypatch.setattr(ops, "_require_gpu", lambda t, name: None)
    monkeypatch.setattr(ops, "_stream", lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(ops, "_stream_handle", lambda: 0)
    for cache in ("_FN", "_NT_FN", "_TN_FN"):                       # entry points cached per process: fresh ones for the stub library
        monkeypatch.setattr(ops, cache, {})
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    return lib


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
def test_engine_launch_sequence_dry_run(stub, name):
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    cfg = config_by_name(name)
    eng = Engine(cfg, "cpu")
    ocfg = O.config_by_name(name)
    eng.params.load_named(O.init_params(ocfg))
    batch = O.synthetic_batch(ocfg, 8, min_len=4)
    out = eng.train_step(batch)
    assert set(out) >= {"loss", "l_loss", "g_loss", "classifier_loss", "classifier_acc"}
    n = stub.calls
    L = cfg.n_layer_v
    assert n.count("medmoe_attn_fwd") == L and n.count("medmoe_attn_bwd") == L
    # text tower on the packed non-padding tokens: one pack, every GEMM / LayerNorm with the device-side row count
    assert n.count("medmoe_text_pack") == 1 and n.count("medmoe_attn_fwd_varlen") == cfg.n_layer_t
    assert n.count("medmoe_gemm_nt_rows") == 4 * cfg.n_layer_t and n.count("medmoe_layernorm_fwd_rows") == 2 * cfg.n_layer_t
    assert n.count("medmoe_text_aggregate_packed") == 1 and n.count("medmoe_text_aggregate") == 0
    n_class = len({(max(1, min(int(v), cfg.max_len)) + 15) // 16 for v in eng.cap_lens.tolist()})      # caption length classes
    # transposed local loss (64 regions: pair3.hip has the instantiation): per class one score GEMM, one forward and one backward pair
    # launch; then the two column-group wgrad-shaped GEMMs; no scale pass (the backward launch takes dL/dsim)
    assert eng.local_t
    assert n.count("medmoe_local_scores_t") == n_class and n.count("medmoe_local_pair3") == 2 * n_class
    assert n.count("medmoe_gemm_tn_cols") == 1 and n.count("medmoe_gemm_tn_gram") == 1       # dC on the pair matrix, dGm from A and the row weights
    assert n.count("medmoe_scale_blocks_ragged") == 0 and n.count("medmoe_adam_step") == 1
    p3 = [i for i, x in enumerate(n) if x == "medmoe_local_pair3"]
    ce = [i for i, x in enumerate(n) if x == "medmoe_ce_strided"]
    assert p3[n_class - 1] < ce[-2] < ce[-1] < p3[n_class]                 # the CE over the sim matrix sits between the forward and backward launches
    # every Linear on the trainable path has exactly one wgrad launch
    assert n.count("medmoe_gemm_tn") == 4 * L + 1 + 8 + 1
    named = eng.params.export_named()
    ref = O.init_params(ocfg)
    for k, v in ref.items():
        if not k.startswith("text."):
            assert torch.equal(named[k].reshape(v.shape), v), k


def test_engine_launch_sequence_region_word_layout(stub, monkeypatch):
    """MEDMOE_LOCAL_PAIR3=0 (and geometries without a pair3 instantiation): the [region][word] kernels + the scale pass."""
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    monkeypatch.setenv("MEDMOE_LOCAL_PAIR3", "0")
    monkeypatch.setenv("MEDMOE_TEXT_VARLEN", "0")
    cfg = config_by_name("tiny")
    eng = Engine(cfg, "cpu")
    ocfg = O.config_by_name("tiny")
    eng.params.load_named(O.init_params(ocfg))
    eng.train_step(O.synthetic_batch(ocfg, 8, min_len=4))
    n = stub.calls
    n_class = len({(max(1, min(int(v), cfg.max_len)) + 15) // 16 for v in eng.cap_lens.tolist()})
    assert not eng.local_t
    assert n.count("medmoe_local_scores_ragged") == n_class and n.count("medmoe_local_pair2_ragged") == n_class
    assert n.count("medmoe_scale_blocks_ragged") == 1 and n.count("medmoe_local_pair3") == 0
    assert n.count("medmoe_attn_fwd") == cfg.n_layer_v + cfg.n_layer_t and n.count("medmoe_text_pack") == 0      # padded text tower


def test_segment_map_matches_oracle():
    from medmoe_amd.engine import VocabTables
    rng = np.random.default_rng(0)
    V, B, T = 60, 64, 14
    vt = VocabTables.synthetic(V, "cpu", n_continuation=20)
    ov = O.Vocab.synthetic(V, n_continuation=20)
    ids = rng.integers(3, V, size=(B, T))
    ids[:, 0] = 1
    for b in range(B):
        if b % 7 == 0:
            continue                      # no [SEP] at all: the last word is dropped
        L = rng.integers(2, T + 1)
        ids[b, L - 1] = 2
        ids[b, L:] = 0
    seg, cap = vt.segment_map(torch.from_numpy(ids))
    seg_ref, n_words, cap_ref = O.segment_map(ids, ov)
    assert np.array_equal(seg.numpy(), seg_ref)
    assert np.array_equal(cap.numpy(), cap_ref)


def test_engine_multi_rank_launch_sequence_dry_run(stub, monkeypatch):
    """world_size 2 path of the engine (gathered global loss + bucketed gradient all-reduce) with the
    collectives replaced by local fakes: every gradient bucket is reduced exactly once, in an order the
    backward has already completed, and the gathered-loss launches see [B, W*B] shapes."""
    import medmoe_amd.dist as D
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    cfg = config_by_name("tiny")
    eng = Engine(cfg, "cpu")
    eng.world, eng.rank, eng.dist = 2, 1, True
    order = []
    Real = D.BucketedAllReduce

    class FakeReducer:
        def __init__(self, flat, bounds):
            self.real = Real(flat, bounds)                              # validates the boundaries
            self.n = self.real.n_buckets
        def ready(self, i):
            assert i not in order
            order.append(i)
        def finish(self):
            assert sorted(order) == list(range(self.n))
    monkeypatch.setattr(D, "BucketedAllReduce", FakeReducer)
    monkeypatch.setattr(D, "gather_embeddings", lambda a, b: (torch.cat([a, a]), torch.cat([b, b])))
    monkeypatch.setattr(D, "scatter_key_grads", lambda d: d[: d.shape[0] // 2].clone())
    monkeypatch.setattr(D, "label_offset", lambda B: B * 1)
    ocfg = O.config_by_name("tiny")
    batch = O.synthetic_batch(ocfg, 8, min_len=4)
    eng.train_step(batch)
    L = cfg.n_layer_v
    assert order[0] == L + 1 and order[-1] == 0 and order[1:-1] == list(range(L, 0, -1))
    assert eng.ws["S"].shape == (8, 16) and eng.ws["d_img_all"].shape == (16, cfg.d_out)
    assert stub.calls.count("medmoe_ce_strided") == 4      # 2 gathered global CE + 2 local (rows, cols)


def test_ragged_layout_tables():
    """Class tables of the ragged local-loss layout: every caption gets pad16(len) private columns, classes are
    contiguous, the chunk map inverts the column map, padding up to Kp is marked -1."""
    from medmoe_amd.engine import ragged_layout
    rng = np.random.default_rng(1)
    for B, T, Tp in ((64, 77, 80), (16, 16, 16), (33, 25, 32), (5, 77, 80)):
        lens = rng.integers(0, T + 5, size=B)                    # includes 0 and > T: clamped to 1..T
        perm, col, ntts, chunk, classes, Kc, Kp = ragged_layout(lens, T, Tp)
        cl = np.clip(lens, 1, T)
        assert np.array_equal(ntts, (cl + 15) // 16) and sorted(perm.tolist()) == list(range(B))
        assert Kc == int((16 * ntts).sum()) and Kp % 64 == 0 and 0 <= Kp - Kc < 64
        used = np.zeros(Kp, bool)
        for i in range(B):
            w = 16 * ntts[i]
            assert w >= cl[i] and not used[col[i]:col[i] + w].any()
            used[col[i]:col[i] + w] = True
            assert (chunk[col[i] // 8:(col[i] + w) // 8] == i).all()
        assert used[:Kc].all() and not used[Kc:].any() and (chunk[Kc // 8:] == -1).all()
        pos = 0
        for ntt, first, n_c, cbase in classes:
            assert first == pos and (ntts[perm[first:first + n_c]] == ntt).all()
            assert np.array_equal(col[perm[first:first + n_c]], cbase + 16 * ntt * np.arange(n_c))
            pos += n_c
        assert pos == B


def test_text_frontend_vocab_hookup_matches_reference_fixture():
    """merged `sents`, cap_lens and the device segment map from a tokenizer vocabulary, against the
    fixture generated by the reference's BertEncoder.aggregate_tokens (tests/golden/bert_aggregate*)."""
    from medmoe_amd.text import merge_sents, cap_lens_from_sents, vocab_tables
    here = os.path.dirname(os.path.abspath(__file__))
    d = np.load(os.path.join(here, "golden", "bert_aggregate.npz"))
    V = d["is_cont"].shape[0]
    words = ["[PAD]", "[CLS]", "[SEP]"] + [f"w{i}" for i in range(3, 30)] + [f"##p{i}" for i in range(30, V)]
    assert [w.startswith("##") for w in words] == d["is_cont"].tolist()
    ref_sents = [line.split(" ") for line in open(os.path.join(here, "golden", "bert_aggregate_sents.txt")).read().splitlines()]
    ids = torch.from_numpy(d["ids"])
    sents = merge_sents(ids, dict(enumerate(words)))
    assert sents == ref_sents
    assert cap_lens_from_sents(sents) == d["cap_lens"].tolist()
    vt = vocab_tables(words, "cpu")
    seg, cap = vt.segment_map(ids)
    assert cap.tolist() == d["cap_lens"].tolist()
    oseg, _, ocap = O.segment_map(d["ids"], O.Vocab(d["is_cont"], d["starts_bracket"]))
    assert np.array_equal(seg.numpy(), oseg) and ocap.tolist() == cap.tolist()
    # no [SEP]: the open word is dropped, as in the reference loop
    assert merge_sents([[1, 5, 31, 7]], words) == [["[CLS]", "w5p31", "[PAD]", "[PAD]"]]


def test_datamodule_mirror_synthetic_loader():
    """UnimedDataModule mirror (constructor arguments of configs/data/unimed.yaml) with the synthetic
    stand-in for the WebDataset shards; collate keys and per-device batch split as in the reference."""
    from src.data.unimed_datamodule import UnimedDataModule
    dm = UnimedDataModule(data_dir="data/", batch_size=8, num_workers=0, pin_memory=False, synthetic_size=32, max_len=25)
    dm.setup(world_size=2)
    assert dm.batch_size_per_device == 4
    batch = next(iter(dm.train_dataloader()))
    assert set(batch) == {"image", "caption", "label"} and len(batch["image"]) == 4 and batch["label"].dtype == torch.long
    im, cap = batch["image"][0], batch["caption"][0]
    assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3 and cap.shape == (25,) and cap[0] == 1
    with pytest.raises(RuntimeError):
        UnimedDataModule(batch_size=7).setup(world_size=2)
