"""CPU-side checks: the C-ABI library loads and exports every symbol include/sntc.h declares, the
host logic (registry, configs, metrics, schedules) mirrors the reference, and the product path
refuses to run without a GPU instead of falling back."""
import json
import math
import os
import re
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from shallow_ntc_amd import _capi
    header = (ROOT / "include" / "sntc.h").read_text()
    declared = set(re.findall(r"\b(sntc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = _capi.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sntc.h but not exported by libsntc_hip.so"
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    assert lib.sntc_version() == 100
    assert isinstance(_capi.last_error(), str)


def test_every_entry_point_cites_the_reference():
    header = (ROOT / "include" / "sntc.h").read_text()
    for cite in ["common/transforms.py", "common/elic.py", "mshyper/models.py", "common/image_utils.py", "common/data_lib.py",
                 "factorized/models.py"]:
        assert cite in header


def test_no_cpu_fallback_in_the_product():
    """Nothing under shallow-ntc_amd/ may import the oracle; without a GPU the ops raise."""
    import torch
    for f in (ROOT / "shallow-ntc_amd").rglob("*.py"):
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
        assert "torch.nn.functional" not in src and "F.conv" not in src, f
    if not torch.cuda.is_available():
        from shallow_ntc_amd import _capi, ops
        from shallow_ntc_amd.mshyper.models import Model
        from shallow_ntc_amd.mshyper import configs
        assert _capi.device_count() == 0
        with pytest.raises(_capi.SntcError):
            Model(**configs.two_layer_syn())
        with pytest.raises(_capi.SntcError):
            ops.ConvPlan("conv", torch.zeros((1, 1, 32, 32)), None, 1)
        with pytest.raises(ValueError):
            ops.pad_reflect(torch.zeros((1, 4, 4, 3)), 8, 8)


def test_registry_matches_reference_names():
    """reference common/transforms.py:383-391."""
    from shallow_ntc_amd.common.transforms import class_builder
    assert set(class_builder) == {
        "BLS2017Analysis", "BLS2017Synthesis", "CNNAnalysis", "CNNSynthesis", "HyperAnalysis", "HyperSynthesis",
        "MBT2018Analysis", "MBT2018Synthesis", "HyperAnalysisSmall", "HyperSynthesisSmall", "ElicAnalysis",
        "ElicSynthesis", "JPEGLikeSynthesis", "TwoLayerSynthesis", "TwoLayerResSynthesis", "JPEGLikeHyperSynthesis"}
    t = class_builder.build("ElicAnalysis", channels=(192, 192, 192, 320))
    assert t.num_params(3) == 7337792 and t.out_channels(3) == 320 and t.out_hw(512, 768) == (32, 48)
    assert class_builder.build("TwoLayerResSynthesis").num_params(320) == 1299003
    assert class_builder.build("TwoLayerSynthesis", channels=(24, 3)).num_params(320) == 1300347
    assert class_builder.build("JPEGLikeSynthesis", kernel_size=18, strides=16).num_params(320) == 311043
    assert class_builder.build("HyperAnalysis", bottleneck_size=320).num_params(320) == 6042560
    assert class_builder.build("HyperSynthesis", bottleneck_size=320).num_params(320) == 9166240
    assert class_builder.build("CNNAnalysis", channels_base=192, output_channels=320, activation_type="gdn").num_params(3) == 3431552
    with pytest.raises(ValueError):
        class_builder.build("ElicAnalysis", channels=(1, 2))
    with pytest.raises(KeyError):
        class_builder.build("NoSuchTransform")


def test_product_and_oracle_inventories_agree():
    from oracle import transforms_np as T
    from shallow_ntc_amd.common.transforms import class_builder
    cases = [("ElicAnalysis", dict(channels=(192, 192, 192, 320)), 3), ("TwoLayerResSynthesis", {}, 320),
             ("BLS2017Analysis", dict(num_filters=256), 3), ("BLS2017Synthesis", dict(num_filters=256), 256),
             ("MBT2018Analysis", dict(channels_base=192, output_channels=320), 3),
             ("MBT2018Synthesis", dict(channels_base=192, output_channels=3), 320),
             ("CNNAnalysis", dict(channels_base=256, output_channels=320), 3),
             ("HyperSynthesis", dict(bottleneck_size=320), 320), ("ElicSynthesis", dict(channels=(192, 160, 128, 3)), 320)]
    for cls, kw, cin in cases:
        a = class_builder.build(cls, **kw).param_shapes(cin)
        okw = dict(kw)
        if cin != 3 or cls.endswith("Synthesis"):
            okw["cin"] = cin
        b = T.build(cls, **okw).param_shapes()
        assert {k: tuple(v) for k, v in a.items()} == {k: tuple(v) for k, v in b.items()}, cls


def test_configs_mirror_reference_files():
    from shallow_ntc_amd.mshyper import configs
    c = configs.two_layer_syn()
    assert c["transform_config"]["analysis"] == dict(cls="ElicAnalysis", channels=(192, 192, 192, 320))
    assert c["transform_config"]["synthesis"]["kernel_sizes"] == (13, 5) and c["transform_config"]["synthesis"]["strides"] == (8, 2)
    assert configs.jpegl()["transform_config"]["synthesis"] == dict(cls="JPEGLikeSynthesis", kernel_size=18, strides=16)
    assert configs.itinf()["latent_config"]["uq"] == dict(method="sga", tau_r=5e-4, tau_ub=0.5, tau_t0=200)
    assert sorted(configs.RD_LAMBDAS) == [0.00125, 0.0025, 0.005, 0.01, 0.02, 0.04, 0.08]
    assert set(configs.CONFIGS) == {"two_layer_syn", "jpegl", "mbt2018", "two_layer_syn2", "bls2017"}


def test_metrics_and_psnr_host_math():
    from shallow_ntc_amd.common import image_utils
    from shallow_ntc_amd.common.train_lib import Metrics
    pub = json.loads((ROOT / "tests" / "golden" / "published_rows.json").read_text())
    for r in pub["2-layer_syn"]:
        mses, psnrs = image_utils.mse_psnr_from_sse(np.array([r["mse"] * 1000.0]), 1000)
        assert abs(psnrs[0] - r["psnr"]) < 2e-4
    m = Metrics.make()
    m.record_scalars(dict(bpp=0.5, psnr=30.0))
    m2 = Metrics.make()
    m2.record_scalars(dict(bpp=1.5, psnr=32.0))
    assert Metrics.merge_metrics([m, m2]).scalars == dict(bpp=1.0, psnr=31.0)
    assert m.scalars_float == dict(bpp=0.5, psnr=30.0)


def test_sga_schedule_and_data_helpers():
    from shallow_ntc_amd.common import data_lib
    from shallow_ntc_amd.common.latent_rvs_utils import sga_schedule_at_step
    assert sga_schedule_at_step(0, 5e-4, 0.5) == 0.5
    assert abs(sga_schedule_at_step(2200, 5e-4, 0.5) - 0.5 * math.exp(-1)) < 1e-12
    img = data_lib.synthetic_images(1, 16, 24, seed=3)
    assert img.dtype == np.uint8 and img.shape == (1, 16, 24, 3)
    np.testing.assert_array_equal(img, data_lib.synthetic_images(1, 16, 24, seed=3))
    x = data_lib.normalize_image(img)
    assert x.dtype == np.float32 and x.min() >= -0.5 and x.max() <= 0.5
    np.testing.assert_array_equal(np.rint(data_lib.unnormalize_image(x)).astype(np.uint8), img)


def test_eval_lib_runname_and_rows():
    """reference common/utils.py:159-169 docstring examples + eval_lib.py:92-102 row schema."""
    from shallow_ntc_amd.common import eval_lib
    from shallow_ntc_amd.common.train_lib import Metrics
    assert list(eval_lib.parse_runname("dir-lamb=2-arch=2_4_8/tau=1.0-step=0-kerasckpt").items()) == \
        [("lamb", "2"), ("arch", "2_4_8"), ("tau", "1.0"), ("step", "0")]
    p = eval_lib.parse_runname("rd-ms2020-latent_depth=320-lmbda=1e-06-dataset=basenji-bpp=0.000-psnr=19.875.npz", True)
    assert p["latent_depth"] == 320 and p["lmbda"] == 1e-06 and p["dataset"] == "basenji" and p["psnr"] == 19.875
    assert eval_lib.parse_runname("k1=13-arch=2_4_8", True)["arch"] == (2, 4, 8)
    m = Metrics.make()
    m.record_scalars(dict(bpp=0.5, psnr=30.0, mse=65.0, rd_loss=0.825))
    rows = eval_lib.results_rows([m, m], "mshyper-rd_lambda=0.005-bottleneck_size=320")
    assert rows[1]["instance_id"] == 1 and rows[0]["rd_lambda"] == 0.005 and rows[0]["bottleneck_size"] == 320
    pub = json.loads((ROOT / "tests" / "golden" / "published_rows.json").read_text())["2-layer_syn"][0]
    assert set(pub) <= set(rows[0]) | {"msssim", "lpips"}            # published rows carry the same keys (+ metrics)


def test_latest_checkpoint_resolution(tmp_path):
    from shallow_ntc_amd.common import eval_lib
    d = tmp_path / "train" / "checkpoints"
    d.mkdir(parents=True)
    for n in (9, 10):
        (d / f"ckpt-{n}.index").write_bytes(b"")
    assert eval_lib.latest_checkpoint(d).name == "ckpt-10"
    (d / "checkpoint").write_text('model_checkpoint_path: "ckpt-9"\nall_model_checkpoint_paths: "ckpt-9"\n')
    assert eval_lib.latest_checkpoint(d).name == "ckpt-9"


def test_png_dataset_pipeline(tmp_path):
    """data_lib.get_dataset over PNG files (reference common/data_lib.py:86-143): ordered full-size evaluation batches,
    shuffled / repeated / randomly cropped training batches, normalisation to [-0.5, 0.5]."""
    from PIL import Image
    from shallow_ntc_amd.common import data_lib
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, size=(40 + 8 * i, 56, 3)).astype(np.uint8) for i in range(3)]
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(tmp_path / f"img{i:02d}.png")
    val = list(data_lib.get_dataset(str(tmp_path / "*.png"), "val", 1, None))
    assert [b.shape for b in val] == [(1, 40, 56, 3), (1, 48, 56, 3), (1, 56, 56, 3)]
    np.testing.assert_array_equal(val[1][0], data_lib.normalize_image(imgs[1]))
    center = list(data_lib.get_dataset(str(tmp_path / "*.png"), "val", 2, 32))
    assert [b.shape for b in center] == [(2, 32, 32, 3), (1, 32, 32, 3)]
    np.testing.assert_array_equal(center[0][0], data_lib.normalize_image(imgs[0][4:36, 12:44]))
    train = data_lib.get_dataset(str(tmp_path / "*.png"), "train", 2, 16, seed=3)
    batches = [next(train) for _ in range(5)]                       # repeats past the 3 files, always full batches
    assert all(b.shape == (2, 16, 16, 3) and b.dtype == np.float32 and -0.5 <= b.min() and b.max() <= 0.5 for b in batches)
    with pytest.raises(RuntimeError):
        data_lib.get_dataset_from_glob(str(tmp_path / "*.jpg"), False, False, False, 1)


def test_learning_rate_schedule_matches_the_reference_closed_form():
    """CompressionSchedule (reference common/schedule.py:121-123,155-176): base * [1, drop] at int(after * total), times
    min(1, (step + 1) / warmup): the FIRST update already has lr > 0 and warm-up also scales the dropped value."""
    from shallow_ntc_amd.mshyper.models import compression_lr
    cfg = dict(learning_rate=2e-4, reduce_lr_after=0.8, reduce_lr_factor=0.1)          # warmup_until default 0.02
    total = 1000                                                                       # warm-up 20 steps, drop at 800
    assert compression_lr(cfg, total, 0) == pytest.approx(2e-4 * 1 / 20)
    assert compression_lr(cfg, total, 9) == pytest.approx(2e-4 * 10 / 20)
    assert compression_lr(cfg, total, 19) == pytest.approx(2e-4)
    assert compression_lr(cfg, total, 20) == pytest.approx(2e-4)
    assert compression_lr(cfg, total, 799) == pytest.approx(2e-4)
    assert compression_lr(cfg, total, 800) == pytest.approx(2e-5)
    # explicit warmup_steps takes precedence; warm-up that outlasts the drop scales the dropped value too
    late = dict(cfg, warmup_steps=1000)
    assert compression_lr(late, total, 899) == pytest.approx(2e-5 * 900 / 1000)
    assert compression_lr(dict(cfg, warmup_steps=0), total, 0) == pytest.approx(2e-4)
    # independent restatement of schedule_at_step for every step of a short run
    for step in range(0, 60):
        want = 1e-3 * (0.5 if step >= int(0.5 * 50) else 1.0) * min(1.0, (step + 1) / 7)
        got = compression_lr(dict(learning_rate=1e-3, reduce_lr_after=0.5, reduce_lr_factor=0.5, warmup_steps=7), 50, step)
        assert got == pytest.approx(want)


def test_bf16x3_layer_rule_is_a_function_of_layer_and_image_geometry():
    """common/_graph.py: which convolutions take the split-precision kernel under Model(precision="bf16x3") -- decided from the
    layer and ONE image's size only (an encoder and a decoder must agree whatever their batching), here for the layers of the
    two_layer_syn model at Kodak size."""
    from shallow_ntc_amd import _capi as capi
    from shallow_ntc_amd.common import _graph as G
    ok = G.s3_geometry_ok
    # hyper-synthesis on 32 x 48 latents: 8 x 12 -> 16 x 24 -> 32 x 48 -> (mu, sigma)
    assert not ok("convT", 2, 320, 8, 12)            # 96 rows per image: fp32 (its split-K / deep-ring instances)
    assert ok("convT", 2, 480, 16, 24)               # 384 rows x 1920 phase columns: 30 tiles per image
    assert ok("convT", 1, 640, 32, 48)               # 1536 rows x 640 columns: 30 tiles
    assert ok("convT", 8, 24, 32, 48)                # the 13x13 / 8 synthesis: 1536 columns
    assert ok("conv", 2, 192, 512, 768) and ok("conv", 2, 320, 128, 192)      # the encoder's 5x5 / 2 layers
    assert not ok("conv", 2, 320, 32, 48)            # 5x5 / 2 320 -> 320 down to 1/32 resolution: 6 tiles per image, slower pre-split
    assert G.s3_rows("convT", 2, 16, 24) == 384 and G.s3_rows("conv", 2, 515, 771) == 258 * 386
    st = G.s3_eligible
    assert st("conv", 192, 96, capi.EPI_STORE) and st("convT", 320, 24, capi.EPI_ADD)
    assert not st("conv", 3, 192, capi.EPI_STORE) and not st("convT", 320, 3, capi.EPI_STORE)      # Cin % 16, Cout % 4
    assert not st("conv", 192, 192, capi.EPI_RES_DIV)                                               # GDN epilogues stay fp32


def test_parity_prose_is_rendered_from_the_committed_reports():
    """No hand-typed parity number in DESIGN.md / README.md: their parity tables and sentence are what tools/parity_tables.py
    renders from profiles/*_e2e_parity*.json (the reports the -m gpu parity tests write on the MI355X)."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "parity_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("k,s,cin,ch,res,h,w", [
    (13, 8, 32, 12, 1, 5, 7), (13, 8, 16, 24, 0, 6, 4), (13, 8, 16, 12, 0, 4, 9), (13, 8, 16, 48, 0, 3, 5), (13, 8, 16, 24, 1, 7, 3),
    (5, 2, 32, 12, 1, 6, 5), (3, 1, 16, 24, 0, 5, 6), (6, 4, 16, 12, 0, 4, 4), (16, 16, 16, 12, 0, 3, 2), (13, 8, 32, 12, 1, 20, 19)])
def test_fused_synthesis_decomposition_on_the_host(k, s, cin, ch, res, h, w):
    """csrc/syn_fused.hip's units, shift lists, slot -> phase map and packed weights, driven through the kernel's loop nest on the
    CPU (sntc_syn_selfcheck: no device involved), reproduce Conv2DTranspose(SAME) (reference common/transforms.py:307-313,
    :331-338): every output pixel is produced by exactly one (unit, slot), and the values agree exactly."""
    import ctypes as C
    from shallow_ntc_amd import _capi as capi
    lib = capi.load()
    assert lib.sntc_syn_supported(k, s, cin, ch, res) == 1
    err = C.c_double(-1.0)
    capi.check(lib.sntc_syn_selfcheck(k, s, cin, ch, res, h, w, 1234, C.byref(err)))
    assert 0.0 <= err.value < 1e-9


def test_fused_synthesis_refuses_what_it_does_not_cover():
    from shallow_ntc_amd import _capi as capi
    lib = capi.load()
    assert lib.sntc_syn_supported(18, 16, 320, 3, 0) == 0       # JPEG-like synthesis: 3 columns per phase
    assert lib.sntc_syn_supported(13, 8, 24, 12, 1) == 0        # input channels not a multiple of 16
    assert lib.sntc_syn_supported(29, 8, 32, 12, 1) == 0        # taps further than one pixel from the aligned source
    assert lib.sntc_syn_supported(13, 8, 320, 6, 1) == 0        # 6 + 6 channels: no kernel instance (6 without residual neither)
    assert lib.sntc_syn_supported(13, 8, 320, 6, 0) == 0
    assert lib.sntc_syn_supported(13, 8, 320, 12, 1) == 1 and lib.sntc_syn_supported(13, 8, 320, 24, 0) == 1


@pytest.mark.parametrize("n_keep", [1, 2, 4, 5])
def test_checkpoint_retention_keeps_the_newest_n(n_keep):
    """tf.train.CheckpointManager(max_to_keep=N) (reference common/train_lib.py:124-126) over a run of saves: after every save the
    newest min(N, saves so far) checkpoints exist, the one just written among them (N >= 4 used to drop checkpoints early: a
    negative slice start)."""
    from shallow_ntc_amd.train import checkpoints_to_keep
    on_disk = {}
    for i, step in enumerate(range(3, 33, 3)):
        keep = checkpoints_to_keep(dict(on_disk), step, n_keep)
        on_disk[step] = float(i)
        on_disk = {k: v for k, v in on_disk.items() if k in keep}
        want = list(range(3, step + 1, 3))[-n_keep:]
        assert keep == want and sorted(on_disk) == want
    # recency, not step number: a stale bundle with a higher step goes first
    assert checkpoints_to_keep({99: 0.0, 6: 5.0, 9: 6.0}, 12, 3) == [6, 9, 12]
    assert checkpoints_to_keep({99: 0.0}, 12, 1) == [12]


@pytest.mark.skipif(not os.environ.get("SNTC_SLOW_TESTS"), reason="recompiles three kernel sources (about four minutes): "
                    "SNTC_SLOW_TESTS=1, or python tools/kernel_resources.py --check, when a kernel source changes")
def test_hot_kernel_register_allocation_is_the_committed_one():
    """The fp32 gather-GEMM instances that carry the decode, the ResidualBlock kernel and the synthesis kernel compile to the
    register / scratch figures committed in profiles/kernel_resources.json: a new mode of the 11-parameter template (or a
    compiler update) that moves them shows up here, not as an unexplained 8 % on the GPU."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_resources.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_static_schedules_environment_switch():
    """SNTC_STATIC_SCHEDULES=1: the process starts with stream-K off (a GPU shared with other processes cannot promise the
    co-residency its hand-offs need; include/sntc.h "Stream-K health") -- read once, when the library is loaded."""
    import os
    import subprocess
    import sys
    code = ("import __graft_entry__ as g; g.load_package(); from shallow_ntc_amd import _capi; "
            "print(int(_capi.load().sntc_conv_get_stream_k()))")
    root = str(Path(__file__).resolve().parent.parent)
    for env_val, want in (("1", "0"), ("", "1"), ("0", "1")):
        env = dict(os.environ, SNTC_STATIC_SCHEDULES=env_val, PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-500:]
        assert out.stdout.strip().splitlines()[-1] == want, (env_val, out.stdout)
