"""End-to-end drive of the reference's step 3 / 4 / 6a / 6b call sequence (StartProcess.py:89-175) on synthetic data."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BASE = "automatic-sem-image-segmentation_amd"


def _write_tiles(d, n, size, rng, mask):
    from PIL import Image
    os.makedirs(d, exist_ok=True)
    for i in range(n):
        if mask:
            a = (rng.random((size, size)) > 0.85).astype(np.uint8) * 255
        else:
            a = (rng.random((size, size)) * 255).astype(np.uint8)
        Image.fromarray(a).save(os.path.join(d, f"{i:03d}.tif"))


def test_cyclegan_then_unet_workflow(tmp_path):
    CG = importlib.import_module(BASE + ".CycleGAN")
    UN = importlib.import_module(BASE + ".UNet_Segmentation")
    rng = np.random.default_rng(0)
    root = str(tmp_path)
    data = os.path.join(root, "2_CycleGAN", "data")
    for sub, mask in (("trainA", False), ("trainB", True), ("testA", False), ("testB", True)):
        _write_tiles(os.path.join(data, sub), 4, 64, rng, mask)
    os.makedirs(os.path.join(root, "2_CycleGAN", "Models"))
    os.makedirs(os.path.join(root, "3_UNet", "Models"))

    # step 3 (StartProcess.py:89-104)
    cg = CG.CycleGAN(root_dir=root, image_shape=(64, 64, 1))
    cg.batch_size, cg.epochs, cg.use_data_loader = 2, 2, True
    cg.label_smoothing_factor, cg.gaussian_noise_value, cg.use_skip_connection = 0.0, 0.0, False
    cg.filters, cg.num_residual_blocks_gen = 4, 2
    model = cg.start_training()
    mdir = os.path.join(cg.model_dir, cg.prefix)
    assert sorted(os.listdir(mdir)) == ["checkpoints_001.keras", "checkpoints_002.keras", "model.keras", "training_log.csv"]
    log = open(os.path.join(mdir, "training_log.csv")).read().strip().split("\n")
    assert log[0].split(";")[0] == "epoch" and len(log) == 3 and len(log[0].split(";")) == 15
    # per-epoch preview sheets (the reference's GANMonitor callback, CycleGAN.py:202,810-905): 2 rows x 4 panels, RGB uint8
    from PIL import Image as _I
    idir = os.path.join(root, "2_CycleGAN", "images", cg.prefix)
    assert sorted(os.listdir(idir)) == ["A-B-A_Epoch_00001.tif", "A-B-A_Epoch_00002.tif", "B-A-B_Epoch_00001.tif", "B-A-B_Epoch_00002.tif"]
    sheet = np.array(_I.open(os.path.join(idir, "A-B-A_Epoch_00002.tif")))
    assert sheet.shape == (2 * 64, 4 * 64, 3) and sheet.dtype == np.uint8 and sheet[:, :64].max() == 255
    assert cg.image_pool_a.batch_size == 2 and cg.image_pool_a.num_imgs == 4 * 2 // 2 * 1 * 2  # 2 images per step, 2 steps/epoch, 2 epochs

    # step 4 (StartProcess.py:107-130): masks -> fake images, loading the saved model like a fresh process would
    cg2 = CG.CycleGAN(root_dir=root, image_shape=(64, 64, 1))
    out_a = os.path.join(root, "2_CycleGAN", "generate_images", "A")
    cg2.run_inference(files=os.path.join(data, "trainB"), output_directory=out_a, source_domain="B", tile_images=False, use_gpu=True)
    assert len(os.listdir(out_a)) == 4
    # tiled inference of a larger image must equal whole-image inference away from tile seams only in shape/type here
    big = os.path.join(root, "big")
    _write_tiles(big, 1, 128, rng, False)
    out_b = os.path.join(root, "out_b")
    cg2.run_inference(files=big, output_directory=out_b, source_domain="A", tile_images=True, use_gpu=True)
    from PIL import Image
    im = np.array(Image.open(os.path.join(out_b, "000.tif")))
    assert im.shape == (128, 128) and im.dtype == np.uint8 and im.max() == 255 and im.min() == 0

    # step 6a (StartProcess.py:149-157): fake images + masks
    un = UN.UNet(root_dir=root, image_dir=out_a, mask_dir=os.path.join(data, "trainB"))
    un.batch_size, un.epochs, un.use_dataloader, un.filters = 2, 2, True, 16
    un.contrast_optimization_range = (0.5, 99.5)
    umodel = un.run_training()
    udir = os.path.join(un.model_dir, un.prefix)
    assert {"Checkpoint_Lowest_Loss.keras", "model.keras", "training_log.csv"} <= set(os.listdir(udir))
    hdr = open(os.path.join(udir, "training_log.csv")).read().split("\n")[0].split(";")
    assert hdr == ["epoch", "acc", "loss", "mae", "val_acc", "val_loss", "val_mae"]

    # step 6b (StartProcess.py:160-175)
    un2 = UN.UNet(root_dir=root, image_dir=out_a, mask_dir=os.path.join(data, "trainB"))
    un2.contrast_optimization_range = (0.5, 99.5)
    out_u = os.path.join(root, "out_unet")
    un2.run_inference(files=big, output_directory=out_u, tile_images=False, threshold=-1, watershed_lines=False, use_gpu=True)
    raw = np.array(Image.open(os.path.join(out_u, "000_raw.tif")))
    lab = np.array(Image.open(os.path.join(out_u, "000.tif")))
    assert raw.dtype == np.float32 and raw.shape == (128, 128) and 0.0 <= raw.min() and raw.max() <= 1.0
    assert set(np.unique(lab)) <= {0, 255}
    # saved-model round trip reproduces the in-memory model bit for bit
    p1 = umodel.predict(np.ascontiguousarray(np.array(Image.open(os.path.join(big, "000.tif")), dtype=np.float32)[None, :, :, None] / 255.0))
    p2 = un2.model.predict(np.ascontiguousarray(np.array(Image.open(os.path.join(big, "000.tif")), dtype=np.float32)[None, :, :, None] / 255.0))
    assert bool((p1 == p2).all())


@pytest.mark.parametrize("storage", ["f16", "bf16"])
def test_trainers_build_their_networks_with_the_requested_activation_storage(tmp_path, storage):
    """`activation_storage` (ACTIVATION_STORAGE of the workflow options): the four CycleGAN networks and the MultiResUNet are built with
    16-bit activation storage, one train step of each leaves finite metrics, and the saved model reloads as an fp32 network."""
    import torch
    CG = importlib.import_module(BASE + ".CycleGAN")
    UN = importlib.import_module(BASE + ".UNet_Segmentation")
    L = importlib.import_module(BASE + "._lib")
    want = L.torch_dtype(storage)
    cg = CG.CycleGAN(root_dir=str(tmp_path), image_shape=(64, 64, 1))
    cg.filters, cg.num_residual_blocks_gen, cg.use_skip_connection, cg.activation_storage = 4, 2, False, storage
    model = cg.create_model()
    assert {n.act_dtype for n in (cg.gen_a, cg.gen_b, cg.disc_a, cg.disc_b)} == {want} and model.act_dtype == want
    g = torch.Generator().manual_seed(0)
    a = torch.rand(2, 64, 64, 1, generator=g) * 2 - 1
    b = torch.rand(2, 64, 64, 1, generator=g) * 2 - 1
    m = model.train_step((a.numpy(), b.numpy()))
    assert all(np.isfinite(float(v)) for v in m.values()), m
    fake = CG.CycleGanModel.to_numpy_array(cg.gen_a(a.to(cg.device), training=False))          # what the preview sheets / inference read
    assert fake.dtype == np.float32 and fake.shape == (2, 64, 64, 1) and np.isfinite(fake).all()
    path = str(tmp_path / "cg.keras")
    model.save(path)
    assert CG.CycleGanModel.load(path, cg.device).act_dtype == torch.float32

    un = UN.UNet(root_dir=str(tmp_path), image_dir=str(tmp_path), mask_dir=str(tmp_path))
    un.filters, un.activation_storage = 16, storage
    um = un.create_model(weighting=4.0)
    assert um.net.act_dtype == want and um.act_dtype == want
    x = torch.rand(2, 64, 64, 1, generator=g)
    y = (torch.rand(2, 64, 64, 1, generator=g) > 0.8).float()
    mu = um.train_step((x.numpy(), y.numpy()))
    assert all(np.isfinite(float(v)) for v in mu.values()), mu
    assert um.predict(x.numpy()).dtype == torch.float32
    E = importlib.import_module(BASE + ".engine")          # device-resident batches, as CycleGanModel.train_step takes them
    on_dev = tuple(E.Act(t.to(un.device).contiguous(), requires_grad=False) for t in (x, y))
    assert all(np.isfinite(float(v)) for v in um.train_step(on_dev).values())
