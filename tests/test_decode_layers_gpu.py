"""The decoder layers of a decode position as ONE persistent launch (option ``decode_layers``; csrc/decode_layers.hip, round 6) against
the launch chain it replaces: bit-identical logits at every position and identical captions, for the shapes the product runs -- batch
sizes that fill no / one / several 40-row blocks per cluster, beams 1 / 5 / 10, both 16-bit types, histories on both sides of the
16-key boundary of the attention phase, the model with labels (BASELINE config 5), a prefix, repeated calls on one model (the
hand-over words carry over from launch to launch)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import synth_images  # noqa: E402


def _model(kind, dtype, v=1000, **kw):
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    model = getattr(M, kind)(v, **kw).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=321))
    return model.cuda().to(dtype)


def _decode(model, images, extra, **kw):
    logits = []
    with torch.no_grad():
        toks, lens = model.generate_batch(images, *extra, logits_hook=lambda i, lg: logits.append(lg.float().clone()), **kw)
    return toks, lens, logits


@pytest.mark.parametrize("n_img,beam,max_len,dtype", [(3, 5, 8, torch.bfloat16), (8, 5, 20, torch.float16), (256, 5, 33, torch.bfloat16),
                                                    (40, 1, 10, torch.float16), (38, 10, 18, torch.float16), (300, 10, 6, torch.bfloat16)])
def test_persistent_layers_equal_the_launch_chain(n_img, beam, max_len, dtype):
    from deephumor_amd import hip
    model = _model("CaptioningTransformer", dtype)
    imgs = synth_images(n_img, seed=4).cuda()
    kw = dict(max_len=min(max_len, 32), beam_size=beam, top_k=max(20, beam), temperature=1.1, seed=9)
    with hip.option_scope(decode_layers=0):
        want = _decode(model, imgs, (), **kw)
    with hip.option_scope(decode_layers=1):
        keys = set()
        with hip.profile() as prof:
            got = _decode(model, imgs, (), **kw)
        keys = set(k.split("[")[0].split("{")[0] for k in prof.summary())
        again = _decode(model, imgs, (), **kw)                          # (a second call: other hand-over counter values, same results)
    assert "dh_decode_layers" in keys                                    # the option reached the kernel
    assert len(got[2]) == len(want[2])
    for i, (a, b) in enumerate(zip(got[2], want[2])):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.equal(again[0], want[0]) and torch.equal(again[1], want[1])


def test_persistent_layers_with_labels_and_a_prefix():
    from deephumor_amd import hip
    model = _model("CaptioningTransformerWithLabels", torch.float16)
    imgs = synth_images(21, seed=6).cuda()
    labels = torch.randint(6, 1000, (21, 3)).cuda()
    cap = torch.randint(6, 1000, (21, 3)).cuda()
    kw = dict(max_len=14, beam_size=10, top_k=50, temperature=1.0, seed=3, caption=cap)
    with hip.option_scope(decode_layers=0):
        want = _decode(model, imgs, (labels,), **kw)
    with hip.option_scope(decode_layers=1):
        got = _decode(model, imgs, (labels,), **kw)
    for a, b in zip(got[2], want[2]):
        assert torch.equal(a, b)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
