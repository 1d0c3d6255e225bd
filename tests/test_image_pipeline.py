"""SURVEY.md §8(f) row F1, image half: the reference's host-side ViltProcessor image path against (CPU) the oracle's restatement of
Pillow's 8-bit bicubic resample + transformers' rescale / normalise / pad, and (GPU) the device pipeline.  Integer / byte work and
a table lookup: everything is compared bit for bit."""
import numpy as np
import pytest
import torch

from oracle import image_oracle as io
from climb_amd.data.image_pipeline import resample_coefficients, vilt_output_size, normalize_table

SHAPES = [(480, 640), (333, 500), (50, 70), (1000, 300), (384, 384), (97, 801), (640, 427), (31, 35), (1200, 1600)]


def _images(shapes, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for i, (h, w) in enumerate(shapes):
        if i % 3 == 0:      # smooth content (gradients + a box) exercises the negative bicubic lobes less than noise does: use both
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
            img[h // 4:h // 2, w // 4:w // 2] = (255, 0, 128)
        else:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out.append(img)
    return out


def _hf_processor():
    mod = pytest.importorskip("transformers.models.vilt.image_processing_pil_vilt")
    return mod.ViltImageProcessorPil()


# ------------------------------------------------------------------------------------------------------------- CPU
def test_oracle_resize_is_pillow_bit_for_bit():
    Image = pytest.importorskip("PIL.Image")
    for img in _images(SHAPES):
        oh, ow = io.vilt_output_size(*img.shape[:2])
        ref = np.array(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
        assert np.array_equal(io.pil_bicubic_resize(img, oh, ow), ref), img.shape
    # up- and down-scaling by awkward ratios, one axis unchanged
    img = _images([(123, 457)], seed=3)[0]
    for oh, ow in [(123, 64), (500, 457), (61, 229), (7, 1000)]:
        ref = np.array(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
        assert np.array_equal(io.pil_bicubic_resize(img, oh, ow), ref), (oh, ow)


def test_oracle_batch_matches_transformers_processor():
    proc = _hf_processor()
    imgs = _images(SHAPES[:6], seed=1)
    from PIL import Image
    ref = proc([Image.fromarray(a) for a in imgs], return_tensors="pt")
    px, pm = io.vilt_image_batch(imgs)
    assert ref["pixel_values"].dtype == torch.float32 and ref["pixel_mask"].dtype == torch.int64
    assert torch.equal(torch.from_numpy(px), ref["pixel_values"])
    assert torch.equal(torch.from_numpy(pm), ref["pixel_mask"])


def test_host_tables_match_oracle():
    """The product's vectorised coefficient / size / normalisation tables are the integers and floats the oracle's scalar
    restatement of Resample.c produces."""
    for n_in, n_out in [(640, 512), (480, 384), (500, 576), (70, 512), (1000, 608), (300, 192), (384, 384), (801, 608), (97, 64), (31, 384),
                        (2000, 640), (35, 416), (1600, 512)]:
        ob, ok, oks = io.resample_coeffs(n_in, n_out)
        pb, pk, pks = resample_coefficients(n_in, n_out)
        assert oks == pks and np.array_equal(ob, pb) and np.array_equal(ok, pk), (n_in, n_out)
    assert np.array_equal(io.normalize_lut(), normalize_table())
    mod = pytest.importorskip("transformers.models.vilt.image_processing_pil_vilt")
    rng = np.random.default_rng(5)
    for _ in range(300):
        h, w = int(rng.integers(16, 3000)), int(rng.integers(16, 3000))
        want = mod.get_resize_output_image_size(np.zeros((3, h, w), dtype=np.uint8), shorter=384, longer=int(1333 / 800 * 384), size_divisor=32,
                                                input_data_format="channels_first")
        assert vilt_output_size(h, w) == tuple(want) == io.vilt_output_size(h, w), (h, w)


def test_pipeline_refuses_cpu():
    from climb_amd.data import DeviceImagePipeline
    with pytest.raises(RuntimeError):
        DeviceImagePipeline(torch.device("cpu"))


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_pipeline_bit_exact_vs_oracle_and_transformers():
    from climb_amd.data import DeviceImagePipeline
    dev = torch.device("cuda:0")
    pipe = DeviceImagePipeline(dev)
    imgs = _images(SHAPES, seed=2)
    out = pipe(imgs)
    px, pm = io.vilt_image_batch(imgs)
    assert out["pixel_values"].shape == px.shape and out["pixel_mask"].dtype == torch.int64
    assert torch.equal(out["pixel_values"].cpu(), torch.from_numpy(px))
    assert torch.equal(out["pixel_mask"].cpu(), torch.from_numpy(pm))
    # the third-party code itself (present on the GPU box as well)
    proc = _hf_processor()
    from PIL import Image
    ref = proc([Image.fromarray(a) for a in imgs], return_tensors="pt")
    assert torch.equal(out["pixel_values"].cpu(), ref["pixel_values"]) and torch.equal(out["pixel_mask"].cpu(), ref["pixel_mask"])
    # PIL inputs, a second batch through the same (cached) tables and staging buffer, single image
    out2 = pipe([Image.fromarray(a) for a in imgs[:3]])
    px2, pm2 = io.vilt_image_batch(imgs[:3])
    assert torch.equal(out2["pixel_values"].cpu(), torch.from_numpy(px2)) and torch.equal(out2["pixel_mask"].cpu(), torch.from_numpy(pm2))
    out3 = pipe(imgs[4:5])
    assert out3["pixel_values"].shape == (1, 3, 384, 384) and bool(out3["pixel_mask"].all())


@pytest.mark.gpu
def test_full_size_batch_roundtrip_properties():
    """64 COCO-sized images (BASELINE batch): every output pixel is one of the 256 table values, padding is exactly zero where the
    mask is zero, and masks are rectangles of the planned sizes."""
    from climb_amd.data import DeviceImagePipeline
    rng = np.random.default_rng(9)
    shapes = [(int(rng.integers(300, 641)), int(rng.integers(300, 641))) for _ in range(64)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    out = DeviceImagePipeline(torch.device("cuda:0"))(imgs)
    pv, pm = out["pixel_values"], out["pixel_mask"]
    lut = torch.from_numpy(normalize_table()).to(pv.device)
    assert bool(torch.isin(pv, torch.cat([lut, lut.new_zeros(1)])).all())
    assert bool((pv * (1 - pm[:, None].float()) == 0).all())
    for b, (h, w) in enumerate(shapes):
        dh, dw = vilt_output_size(h, w)
        assert int(pm[b].sum()) == dh * dw and bool(pm[b, :dh, :dw].all())
    # spot-check three images against the oracle
    for b in (0, 31, 63):
        dh, dw = vilt_output_size(*shapes[b])
        want = io.normalize_lut()[io.pil_bicubic_resize(imgs[b], dh, dw)].transpose(2, 0, 1)
        assert torch.equal(pv[b, :, :dh, :dw].cpu(), torch.from_numpy(want))


@pytest.mark.gpu
def test_process_inputs_equals_vilt_processor(tmp_path):
    """The drop-in point itself (REF/modeling/vilt.py:83-96): `process_inputs(images, texts)` with PIL images and strings returns the
    five tensors ViltProcessor returns, bit for bit, with the image half produced on the device."""
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.modeling import create_continual_learner_map
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + "what is the man holding a red umbrella dog on left ? color ##s two".split()
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    from tests.synth_data import make_tokenizer
    tok = make_tokenizer(str(tmp_path / "vocab.txt"), fast=True)          # (transformers 5.x ignores `vocab_file=`: every word would be [UNK])
    proc = transformers.ViltProcessor(image_processor=_hf_processor(), tokenizer=tok)
    dev = torch.device("cuda:0")
    model = create_continual_learner_map["vilt"](model_name_or_path="random-init:3", ordered_cl_tasks=["vqa"], model_config=model_configs["vilt"],
                                                 task_configs=task_configs, device=dev)
    enc_mod = model.get_encoder()
    enc_mod.processor = proc
    imgs = [Image.fromarray(a) for a in _images([(480, 640), (375, 500), (640, 480)], seed=4)]
    texts = ["what is the man holding ?", "two dogs", "what color is the umbrella on the left ?"]
    got = enc_mod.process_inputs(imgs, texts)
    want = proc(images=imgs, text=texts, max_length=40, padding=True, truncation=True, return_tensors="pt")
    assert set(got) == set(want.keys())
    for k in want.keys():
        assert got[k].device.type == "cuda" and got[k].dtype == want[k].dtype and torch.equal(got[k].cpu(), want[k]), k
    # and the encodings run through the model (variable-resolution path, row F2)
    model.eval()
    with torch.no_grad():
        pooled, logits = model(task_key="vqa", images=imgs, texts=texts)
    assert pooled.shape == (3, 768) and logits.shape == (3, 3129) and bool(torch.isfinite(logits).all())
