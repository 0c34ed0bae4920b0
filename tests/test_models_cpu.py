"""CPU-only host-logic tests of the deephumor.models mirror: constructor signatures, state-dict key
layout recorded from the reference, checkpoint round trip, loud failure without a GPU."""
import inspect

import pytest
import torch

import deephumor_amd.models as M
from helpers import KINDS, meta, synthetic_sd, synth_images


@pytest.mark.parametrize("kind", KINDS)
def test_state_dict_layout_matches_reference(kind):
    rec = meta()["models"][kind]
    hp = rec["hp"] or {"num_tokens": 1000}
    model = getattr(M, kind)(**hp)
    sd = model.state_dict()
    assert sorted(sd) == sorted(rec["keys"])
    for k, shape in rec["keys"].items():
        assert list(sd[k].shape) == shape, k
    if rec["hp"]:
        assert model._hp == rec["hp"]
    trunk = model.encoder.resnet if hasattr(model.encoder, "resnet") else model.encoder.image_encoder.resnet
    assert all(not p.requires_grad for p in trunk.parameters())            # encoders.py:35-36
    enc = model.encoder if hasattr(model.encoder, "resnet") else model.encoder.image_encoder
    assert all(p.requires_grad for p in enc.linear.parameters())           # the embedding head stays trainable (:42-43)


def test_signatures_follow_the_reference():
    def names(f):
        return list(inspect.signature(f).parameters)

    assert names(M.ImageEncoder.__init__)[1:] == ["emb_dim", "dropout", "spatial_features"]
    assert names(M.LSTMDecoder.__init__)[1:] == ["num_tokens", "emb_dim", "hidden_size", "num_layers", "dropout", "embedding"]
    assert names(M.LSTMDecoder.generate)[1:8] == ["image_emb", "caption", "max_len", "temperature", "beam_size", "top_k", "eos_index"]
    assert names(M.TransformerDecoder.__init__)[1:] == ["num_tokens", "hid_dim", "n_layers", "n_heads", "pf_dim", "dropout", "pad_index", "max_len"]
    assert names(M.TransformerDecoder.forward)[1:4] == ["x", "enc_out", "start_emb"]
    extra = [p for p in inspect.signature(M.TransformerDecoder.forward).parameters.values()][4:]
    assert all(p.kind is inspect.Parameter.KEYWORD_ONLY for p in extra)        # extensions never shift the positionals
    assert names(M.TransformerDecoder.generate)[1:9] == ["start_emb", "enc_out", "caption", "max_len", "temperature", "beam_size", "top_k", "eos_index"]
    assert names(M.SelfAttentionTransformerDecoder.generate)[1:8] == ["start_emb", "caption", "max_len", "temperature", "beam_size", "top_k", "eos_index"]
    assert names(M.CaptioningLSTM.__init__)[1:] == ["num_tokens", "emb_dim", "hidden_size", "num_layers", "enc_dropout", "dec_dropout"]
    assert names(M.CaptioningLSTM.generate)[1:8] == ["image", "caption", "max_len", "temperature", "beam_size", "top_k", "eos_index"]
    assert names(M.CaptioningLSTMWithLabels.generate)[1:3] == ["image", "label"]
    assert names(M.CaptioningLSTMWithLabels.forward)[1:] == ["images", "captions", "lengths", "labels"]
    assert names(M.CaptioningTransformer.__init__)[1:] == ["num_tokens", "hid_dim", "n_layers", "n_heads", "pf_dim", "enc_dropout", "dec_dropout", "pad_index", "max_len"]
    assert names(M.BeamSearchHelper.__init__)[1:7] == ["temperature", "beam_size", "top_k", "unk_index", "eos_index", "device"]
    d = inspect.signature(M.CaptioningLSTM.generate).parameters
    assert (d["max_len"].default, d["temperature"].default, d["beam_size"].default, d["top_k"].default, d["eos_index"].default) == (25, 1.0, 10, 50, 3)


def test_checkpoint_round_trip(tmp_path):
    sd, hp = synthetic_sd("CaptioningLSTMWithLabels")
    model = M.CaptioningLSTMWithLabels(**hp)
    model.load_state_dict(sd)
    path = str(tmp_path / "m.pth")
    model.save(path)
    ckpt = torch.load(path, map_location="cpu")
    assert set(ckpt) == {"model", "hp"} and ckpt["hp"] == hp
    again = M.CaptioningLSTMWithLabels.from_pretrained(path)
    assert again._hp == hp
    for k, v in model.state_dict().items():
        assert torch.equal(v, again.state_dict()[k])
    # shared embedding (caption_models.py:125)
    assert again.decoder.embedding is again.encoder.label_encoder.embedding
    sd, hp = synthetic_sd("CaptioningTransformer")
    model = M.CaptioningTransformer(**hp)
    model.load_state_dict(sd)
    assert float(model.decoder.scale) == pytest.approx(512 ** 0.5)
    assert float(model.decoder.layers[0].self_attn.scale) == 8.0


def test_beam_size_assertion_and_dead_encoder():
    with pytest.raises(AssertionError):
        M.BeamSearchHelper(beam_size=10, top_k=5, device="cpu")
    with pytest.raises(NotImplementedError):
        M.TransformerEncoder()


def test_no_cpu_fallback():
    model = M.CaptioningLSTM(50).eval()
    with pytest.raises(RuntimeError, match="no CPU"):
        model(synth_images(1, size=32), torch.zeros(1, 3, dtype=torch.long))
    model.train()
    with pytest.raises(RuntimeError, match="eval"):
        model(synth_images(1, size=32), torch.zeros(1, 3, dtype=torch.long))


def test_product_never_imports_the_oracle():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "deephumor_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                for line in src.splitlines():
                    if line.lstrip().startswith(("import ", "from ")):
                        assert "oracle" not in line, (f, line)


def test_torch_cpu_stream_replica_is_bit_equal_to_torch():
    """rng="torch" fills a batch's noise from a numpy replica of the seeded torch CPU generator's ``exponential_`` stream (thread
    parallel over the images): value for value what torch itself draws, across consecutive fills of different lengths."""
    from deephumor_amd.models.beam import TorchRngNoise, _TorchCpuStream
    assert _TorchCpuStream.self_check()
    for seed in (0, 700, 955, 2 ** 33 + 5):
        g, mine = torch.Generator().manual_seed(seed), _TorchCpuStream(seed)
        for n in (5, 5 * 36541, 24, 300000):
            assert bool((torch.empty(n).exponential_(1, generator=g).numpy() == mine.exponential(n)).all()), (seed, n)

    class _H:
        beam_size = 3
        done = torch.zeros(8, dtype=torch.uint8)
        _ended = torch.tensor([0, 1, 0] * 8, dtype=torch.uint8)
    src = TorchRngNoise(100, 8, img0=2)
    src.attach(_H())
    gens = [torch.Generator().manual_seed(102 + i) for i in range(8)]
    row = src("row", 0, (24, 501))
    want = torch.cat([torch.empty(3 * 501).exponential_(1, generator=g).view(3, 501) for g in gens])
    assert torch.equal(row, want)
    cand = src("cand", 0, (8, 9))
    for i, g in enumerate(gens):                  # 3 + 1 + 3 candidates (beam.py:72-101), the rest of the row untouched
        assert torch.equal(cand[i, :7], torch.empty(7).exponential_(1, generator=g)) and bool((cand[i, 7:] == 1).all())


def test_id_and_length_checks_raise_like_torch():
    """beam.check_ids / check_lengths (host logic in front of the kernels): nn.Embedding's IndexError for ids outside [0, n),
    pack_padded_sequence's RuntimeErrors for lengths <= 0 or beyond the padded sequence; empty and None inputs pass."""
    import pytest
    import torch
    from deephumor_amd.models.beam import check_ids, check_lengths
    emb = torch.nn.Embedding(10, 4)
    for bad in (torch.tensor([[1, 10]]), torch.tensor([[-1, 3]]), torch.tensor([[2 ** 31 + 5]])):
        with pytest.raises(IndexError, match="index out of range"):
            check_ids(bad, 10)
        with pytest.raises(IndexError):                    # what the reference's lookup does with the same ids
            emb(bad)
    check_ids(torch.tensor([[0, 9]]), 10)
    check_ids(torch.zeros((3, 0), dtype=torch.long), 10)
    check_ids(None, 10)
    x = torch.zeros(2, 5, 3)
    for lens, pat in ((torch.tensor([5, 0]), "greater than 0"), (torch.tensor([6, 2]), "sequence length")):
        with pytest.raises(RuntimeError, match=pat):
            check_lengths(lens, 5)
        with pytest.raises(RuntimeError):                  # the reference: at pack time (<= 0) or inside the LSTM call (too long)
            torch.nn.LSTM(3, 4, batch_first=True)(torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False))
    check_lengths(torch.tensor([5, 1]), 5)
