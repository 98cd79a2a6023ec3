"""Frame preprocessing (SURVEY.md par. 8f.1): oracle known-answer cases (CPU) and the HIP kernel against
the oracle (GPU)."""
import pytest
import torch

from oracle.preprocess_oracle import MEAN, STD, image_transform


def test_oracle_known_answers():
    # constant BGR image: every output pixel is ((value/255) - mean_c) / std_c with the channels swapped
    img = torch.zeros(240, 320, 3, dtype=torch.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 10, 128, 250              # B, G, R
    out = image_transform(img)
    assert out.shape == (3, 224, 224)
    for c, v in enumerate((250, 128, 10)):                            # RGB order on the output
        want = (v / 255.0 - MEAN[c]) / STD[c]
        assert torch.allclose(out[c], torch.full((224, 224), want), atol=2e-5)
    # 224x224 input: Resize and CenterCrop are identities -> exact per-pixel formula
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (224, 224, 3), dtype=torch.uint8, generator=g)
    out = image_transform(img)
    want = (img.permute(2, 0, 1).float()[[2, 1, 0]] / 255.0 - torch.tensor(MEAN)[:, None, None]) / torch.tensor(STD)[:, None, None]
    assert torch.equal(out, want)
    # centre crop of a wide frame keeps the middle columns (no resize along H when H == 224)
    img = torch.zeros(224, 448, 3, dtype=torch.uint8)
    img[:, 112:336] = 255
    assert torch.allclose(image_transform(img)[0], torch.full((224, 224), (1.0 - MEAN[0]) / STD[0]), atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(240, 320), (224, 224), (480, 360), (720, 1280), (225, 224), (300, 301)])
def test_kernel_vs_oracle(H, W):
    from gitcap.preprocess import preprocess_frames
    g = torch.Generator().manual_seed(H * 1000 + W)
    frames = torch.randint(0, 256, (2, 3, H, W, 3), dtype=torch.uint8, generator=g)      # [B, F, H, W, 3]
    out = preprocess_frames(frames).cpu()
    assert out.shape == (2, 3, 3, 224, 224)
    want = torch.stack([torch.stack([image_transform(f) for f in clip]) for clip in frames])
    assert (out - want).abs().max() < 2e-4, float((out - want).abs().max())


@pytest.mark.gpu
def test_raw_frames_to_caption():
    """uint8 camera frames -> device preprocessing -> greedy caption, against the same pipeline with the
    oracle transform on the host (what src/real_time_inference.py does)."""
    from gitcap.config import git_base
    from gitcap.model import GitCaptioner
    from gitcap.preprocess import preprocess_frames
    from gitcap.weights import synthetic_weights
    cfg = git_base(6)
    m = GitCaptioner(cfg, synthetic_weights(cfg, 0), max_batch=1, max_text_len=12)
    g = torch.Generator().manual_seed(3)
    raw = torch.randint(0, 256, (1, 6, 480, 640, 3), dtype=torch.uint8, generator=g)
    dev_in = preprocess_frames(raw)
    host_in = torch.stack([image_transform(f) for f in raw[0]])[None]
    assert (dev_in.cpu() - host_in).abs().max() < 2e-4
    a = m.greedy_decode(dev_in, max_len=10, stop="never").cpu()
    b = m.greedy_decode(host_in, max_len=10, stop="never")
    assert a.shape == (1, 11)
    # the two inputs differ by <= 2e-4 per pixel: the captions agree up to the first near-tie, and every token the
    # device-preprocessed path chose is (within 0.05 of) the arg-max under the host-preprocessed frames
    lg = m(host_in, a[:, :-1])
    gap = lg.max(-1).values - lg.gather(2, a[:, 1:, None].to(lg.device)).squeeze(-1)
    assert float(gap.max()) < 0.05, gap
    first = int((a != b).any(0).float().argmax()) if bool((a != b).any()) else a.shape[1]
    assert first == a.shape[1] or float(gap[0, first - 1]) > 0, (a, b)
    with pytest.raises(ValueError):
        preprocess_frames(torch.zeros(4, 4, 3))
    # raw uint8 frames straight into the model: the transform fused with the patch gather (gitcap_greedy_raw /
    # gitcap_encode_raw) gives BITWISE the result of preprocess_frames followed by the fp32 entry points
    c = m.greedy_decode(raw, max_len=10, stop="never").cpu()
    assert torch.equal(c, a)
    _, v_raw = m.forward_image_enc(raw)
    _, v_two = m.forward_image_enc(dev_in)
    assert torch.equal(v_raw, v_two)
    small = torch.randint(0, 256, (1, 6, 100, 300, 3), dtype=torch.uint8, generator=g)      # shorter side < crop: upsampled
    assert torch.equal(m.greedy_decode(small, max_len=4, stop="never").cpu(),
                       m.greedy_decode(preprocess_frames(small), max_len=4, stop="never").cpu())
