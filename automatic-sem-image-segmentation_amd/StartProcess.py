"""Workflow driver with the option surface of the reference's ``StartProcess.py`` (Releases/Version 1.2.0/StartProcess.py:14-43 the
options, :55-175 the steps, :178-221 the step sequence), executing on libsemseg_hip.so.

    python -m automatic-sem-image-segmentation_amd.StartProcess --root /data/run1                       # all steps, reference defaults
    python StartProcess.py --root /data/run1 --steps 3,4,5 --set CYCLEGAN_EPOCHS=10 --set TILE_SIZE_W=512 --set TILE_SIZE_H=512
    python -m torch.distributed.run --nproc-per-node 8 StartProcess.py --root /data/run1 --steps 3      # tile data parallel (RCCL)

* ``WorkflowOptions`` carries every constant the reference defines at module level, under the SAME NAMES and with the same
  defaults, so a settings block copied from an edited reference ``StartProcess.py`` is valid input (``--set NAME=value`` or
  ``WorkflowOptions(NAME=value)``).  Options that only make sense for TensorFlow / CUDA device selection (ALLOW_MEMORY_GROWTH,
  USE_GPUS_NO) are accepted and passed on; device placement is one process per GPU (``dist.init_from_env``).
* ``Workflow.step_0 .. step_6b`` are the reference's ``start_step_*`` functions: the same calls with the same keyword values.
* The reference runs each step in a freshly spawned process because TensorFlow does not give GPU memory back
  (StartProcess.py:45-47,178-221).  Here memory is owned by torch's caching allocator and released between steps
  (``torch.cuda.empty_cache``), so the steps run in-process by default; ``--spawn`` reproduces the process-per-step behaviour.
"""
import argparse
import dataclasses
import os
import sys
from datetime import datetime

if __package__ in (None, ""):          # run as a script from inside the package directory, like the reference's StartProcess.py
    import importlib
    _PKG_DIR = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(_PKG_DIR))
    _PKG = importlib.import_module(os.path.basename(_PKG_DIR))
    WassersteinGAN = importlib.import_module(_PKG.__name__ + ".WassersteinGAN")
    CycleGAN = importlib.import_module(_PKG.__name__ + ".CycleGAN")
    UNet_Segmentation = importlib.import_module(_PKG.__name__ + ".UNet_Segmentation")
    HelperFunctions = importlib.import_module(_PKG.__name__ + ".HelperFunctions")
    dist = importlib.import_module(_PKG.__name__ + ".dist")
else:
    from . import CycleGAN, HelperFunctions, UNet_Segmentation, WassersteinGAN, dist


@dataclasses.dataclass
class WorkflowOptions:
    """StartProcess.py:14-43, same names, same defaults (directories default to sub-directories of ROOT_DIR as there)."""
    # General setup
    ROOT_DIR: str = os.path.abspath("./")
    INPUT_DIR_MASKS: str = None                      # <ROOT_DIR>/Input_Masks (read by WGAN as root_dir/Input_Masks)
    INPUT_DIR_IMAGES: str = None                     # <ROOT_DIR>/Input_Images
    OUTPUT_DIR_CYCLEGAN: str = None                  # <ROOT_DIR>/Output_Masks_CycleGAN
    OUTPUT_DIR_UNET: str = None                      # <ROOT_DIR>/Output_Masks_UNet
    TILE_SIZE_W: int = 384
    TILE_SIZE_H: int = 384
    NUM_SIMULATED_MASKS: int = 1000
    RUN_INFERENCE_ON_WHOLE_IMAGE: bool = True
    DARK_BACKGROUND: bool = True
    # GPU options
    USE_GPUS_NO: tuple = (0,)
    USE_GPU_FOR_WHOLE_IMAGE_INFERENCE: bool = False
    ALLOW_MEMORY_GROWTH: bool = True
    # Training options
    WGAN_BATCH_SIZE: int = 64
    WGAN_EPOCHS: int = 1000
    MAX_PARTICLE_OVERLAP: float = 0.5
    CYCLEGAN_BATCH_SIZE: int = 5
    CYCLEGAN_EPOCHS: int = 50
    CYCLEGAN_USE_SKIPS: bool = False
    CYCLEGAN_FILTERS: int = 64
    GAUSSIAN_BLUR_AMOUNT: float = 0.0
    UNET_BATCH_SIZE: int = 5
    UNET_EPOCHS: int = 50
    UNET_CONTRAST_OPTIMIZATION_RANGE: tuple = (0.5, 99.5)
    UNET_FILTERS: int = 16
    USE_DATALOADER: bool = True
    # step-2 knobs the reference hard-codes inside start_step_2 (StartProcess.py:72-85); exposed so that small runs are possible
    MIN_NO_OF_PARTICLES: int = 100
    MAX_NO_OF_PARTICLES: int = 150
    WGAN_NOISE_DIM: int = 128
    # not in the reference: activation storage of the CycleGAN and MultiResUNet training steps, 'f32' | 'bf16' | 'f16'
    ACTIVATION_STORAGE: str = os.environ.get("SS_ACT_DTYPE", "f32")

    _DERIVED_DIRS = (("INPUT_DIR_MASKS", "Input_Masks"), ("INPUT_DIR_IMAGES", "Input_Images"),
                     ("OUTPUT_DIR_CYCLEGAN", "Output_Masks_CycleGAN"), ("OUTPUT_DIR_UNET", "Output_Masks_UNet"))

    def __post_init__(self):
        self.ROOT_DIR = os.path.abspath(self.ROOT_DIR)
        # directories the caller did not name follow ROOT_DIR -- also when ROOT_DIR changes later (`set`): remember which they are
        defaulted = self.__dict__.setdefault("_defaulted", set())
        for name, sub in self._DERIVED_DIRS:
            if getattr(self, name) is None or name in defaulted:
                defaulted.add(name)
                setattr(self, name, os.path.join(self.ROOT_DIR, sub))
        if not isinstance(self.USE_GPUS_NO, (list, tuple)):
            self.USE_GPUS_NO = (self.USE_GPUS_NO,)

    @property
    def use_gpu_for_inference(self):          # StartProcess.py:49
        return (not self.RUN_INFERENCE_ON_WHOLE_IMAGE) or (self.USE_GPU_FOR_WHOLE_IMAGE_INFERENCE and self.RUN_INFERENCE_ON_WHOLE_IMAGE)

    def set(self, name, text):
        """``--set NAME=value``: value parsed by the type of the default (bool: 1/0/true/false; tuples: comma separated)."""
        fields = {f.name: f for f in dataclasses.fields(self)}
        if name not in fields:
            raise KeyError(f"unknown option {name}; known: {', '.join(sorted(fields))}")
        cur = getattr(self, name)
        if isinstance(cur, bool):
            val = text.strip().lower() in ("1", "true", "yes", "on")
        elif isinstance(cur, int):
            val = int(text)
        elif isinstance(cur, float):
            val = float(text)
        elif isinstance(cur, (tuple, list)):
            val = tuple(type(cur[0])(v) for v in text.split(",")) if cur else tuple(text.split(","))
        else:
            val = text
        setattr(self, name, val)
        self.__dict__.setdefault("_defaulted", set()).discard(name)          # an explicitly set directory no longer follows ROOT_DIR
        self.__post_init__()


class Workflow:
    """The eight steps of StartProcess.py (:55-175) as methods; ``run`` = its ``__main__`` block (:178-221)."""

    ORDER = ("0", "1", "2", "3", "4", "5", "6a", "6b")

    def __init__(self, options=None, **overrides):
        self.o = options if options is not None else WorkflowOptions(**overrides)
        # host-side torch ops (conversions, the loaders' arrays): no more OpenMP threads than this process may use (a container's CPU
        # quota is not what torch sees: 256 threads on a 16-CPU quota only get throttled)
        import torch
        torch.set_num_threads(max(1, min(torch.get_num_threads(), HelperFunctions.usable_cores())))

    def _cyclegan(self):
        o = self.o
        cg = CycleGAN.CycleGAN(root_dir=o.ROOT_DIR, image_shape=(o.TILE_SIZE_H, o.TILE_SIZE_W, 1), allow_memory_growth=o.ALLOW_MEMORY_GROWTH,
                               use_gpus_no=o.USE_GPUS_NO)
        cg.use_skip_connection = o.CYCLEGAN_USE_SKIPS
        cg.filters = o.CYCLEGAN_FILTERS
        cg.use_binary_crossentropy = False
        cg.use_resize_convolution = False
        cg.activation_storage = o.ACTIVATION_STORAGE
        return cg

    def _unet(self):
        o = self.o
        gen = os.path.join(o.ROOT_DIR, '2_CycleGAN', 'generate_images')
        un = UNet_Segmentation.UNet(root_dir=o.ROOT_DIR, image_dir=os.path.join(gen, 'A'), mask_dir=os.path.join(gen, 'Synthetic_Masks_Filtered'),
                                    allow_memory_growth=o.ALLOW_MEMORY_GROWTH, use_gpus_no=o.USE_GPUS_NO)
        un.use_dataloader = o.USE_DATALOADER
        un.filters = o.UNET_FILTERS
        un.contrast_optimization_range = o.UNET_CONTRAST_OPTIMIZATION_RANGE
        un.activation_storage = o.ACTIVATION_STORAGE
        return un

    def step_0(self):
        """Directories + tiling of the input images into 2_CycleGAN/data/trainA (StartProcess.py:55-58)."""
        o = self.o
        HelperFunctions.initialize_directories(root_dir=o.ROOT_DIR, output_dir_cyclegan=o.OUTPUT_DIR_CYCLEGAN, output_dir_unet=o.OUTPUT_DIR_UNET)
        HelperFunctions.prepare_images_cycle_gan(root_dir=o.ROOT_DIR, input_dir_images=o.INPUT_DIR_IMAGES, tile_size_w=o.TILE_SIZE_W,
                                                 tile_size_h=o.TILE_SIZE_H, num_simulated_masks=o.NUM_SIMULATED_MASKS,
                                                 dark_background=o.DARK_BACKGROUND)

    def step_1(self):
        """WGAN-GP on the example particle masks (StartProcess.py:61-67)."""
        o = self.o
        wgan = WassersteinGAN.WGAN(root_dir=o.ROOT_DIR, allow_memory_growth=o.ALLOW_MEMORY_GROWTH, use_gpus_no=o.USE_GPUS_NO)
        wgan.batch_size, wgan.epochs, wgan.n_z = o.WGAN_BATCH_SIZE, o.WGAN_EPOCHS, o.WGAN_NOISE_DIM
        return wgan.start_training()

    def step_2(self):
        """Simulated masks -> 2_CycleGAN/data/trainB (StartProcess.py:70-87)."""
        o = self.o
        num_masks = max(o.NUM_SIMULATED_MASKS, len(os.listdir(os.path.join(o.ROOT_DIR, '2_CycleGAN', 'data', 'trainA'))))
        wgan = WassersteinGAN.WGAN(root_dir=o.ROOT_DIR, allow_memory_growth=o.ALLOW_MEMORY_GROWTH, use_gpus_no=o.USE_GPUS_NO)
        wgan.n_z = o.WGAN_NOISE_DIM
        wgan.simulate_masks(no_of_images=num_masks, min_no_of_particles=o.MIN_NO_OF_PARTICLES, max_no_of_particles=o.MAX_NO_OF_PARTICLES,
                            use_perlin_noise=True, perlin_noise_threshold=0.5, perlin_noise_frequency=4, use_normal_distribution=True,
                            use_random_rotation='DISABLE', grid_type='DISABLE', max_overlap=o.MAX_PARTICLE_OVERLAP,
                            img_width=o.TILE_SIZE_W, img_height=o.TILE_SIZE_H)

    def step_3(self):
        """CycleGAN training (StartProcess.py:90-105)."""
        o = self.o
        cg = self._cyclegan()
        cg.batch_size, cg.epochs, cg.use_data_loader = o.CYCLEGAN_BATCH_SIZE, o.CYCLEGAN_EPOCHS, o.USE_DATALOADER
        cg.label_smoothing_factor = 0.0
        cg.gaussian_noise_value = 0.0
        cg.lambda_identity_a = cg.lambda_identity_b = 0.5
        return cg.start_training()

    def step_4(self):
        """Fake SEM images from the simulated masks + CycleGAN segmentation of the real images (StartProcess.py:108-131)."""
        o = self.o
        cg = self._cyclegan()
        gen = os.path.join(o.ROOT_DIR, '2_CycleGAN', 'generate_images')
        cg.run_inference(files=os.path.join(o.ROOT_DIR, '2_CycleGAN', 'data', 'trainB'), output_directory=os.path.join(gen, 'A'),
                         source_domain='B', tile_images=False, use_gpu=True)
        cg.image_shape = (o.TILE_SIZE_W, o.TILE_SIZE_H)
        cg.run_inference(files=o.INPUT_DIR_IMAGES, output_directory=os.path.join(gen, 'B'), source_domain='A',
                         tile_images=not o.RUN_INFERENCE_ON_WHOLE_IMAGE, min_overlap=2, manage_overlap_mode=2, use_gpu=o.use_gpu_for_inference)

    def step_5(self):
        """Filter the simulated masks by what the CycleGAN rendered; post-process its segmentations (StartProcess.py:134-147)."""
        o = self.o
        gen = os.path.join(o.ROOT_DIR, '2_CycleGAN', 'generate_images')
        HelperFunctions.filter_gan_masks(img_path=os.path.join(gen, 'A'), msk_path=os.path.join(o.ROOT_DIR, '2_CycleGAN', 'data', 'trainB'),
                                         out_path=os.path.join(gen, 'Synthetic_Masks_Filtered'), gaussian_blur_amount=o.GAUSSIAN_BLUR_AMOUNT,
                                         do_watershed_and_four_connectivity=False, dark_background=o.DARK_BACKGROUND)
        HelperFunctions.filter_gan_masks(img_path=o.INPUT_DIR_IMAGES, msk_path=os.path.join(gen, 'B'), out_path=o.OUTPUT_DIR_CYCLEGAN,
                                         do_watershed_and_four_connectivity=True, dark_background=o.DARK_BACKGROUND)

    def step_6a(self):
        """MultiResUNet training on (fake image, filtered mask) pairs (StartProcess.py:150-157)."""
        o = self.o
        un = self._unet()
        un.batch_size, un.epochs = o.UNET_BATCH_SIZE, o.UNET_EPOCHS
        return un.run_training()

    def step_6b(self):
        """UNet segmentation of the real images (StartProcess.py:160-175)."""
        o = self.o
        un = self._unet()
        un.image_shape = (o.TILE_SIZE_W, o.TILE_SIZE_H)
        un.run_inference(files=o.INPUT_DIR_IMAGES, output_directory=o.OUTPUT_DIR_UNET, tile_images=not o.RUN_INFERENCE_ON_WHOLE_IMAGE,
                         threshold=-1, watershed_lines=True, min_distance=9, min_overlap=2, manage_overlap_mode=2,
                         use_gpu=o.use_gpu_for_inference)

    TITLES = {"0": "Step 0: Initializing Directories and Preparing Images...", "1": "Step 1: Training WGAN...",
              "2": "Step 2: Simulating fake masks...", "3": "Step 3: Training CycleGAN...",
              "4": "Step 4: Generating fake training images and segmenting real images with CycleGAN...",
              "5": "Step 5: Postprocessing CycleGAN Output images...", "6a": "Step 6.a: Train MultiRes UNet...",
              "6b": "Step 6.b: Segment real images with UNet"}
    HOST_ONLY = ("0", "5")          # steps without device work: rank 0 alone runs them under torch.distributed

    def run_step(self, key):
        import torch
        if dist.rank() == 0:
            print(self.TITLES[key], flush=True)
        if key in self.HOST_ONLY or key in ("2", "4", "6b"):          # file-producing steps: one writer, working alone --
            if dist.rank() == 0:                                        # the models it builds must not broadcast (dist.solo)
                with dist.solo():
                    getattr(self, "step_" + key)()
        else:
            getattr(self, "step_" + key)()
        if dist.world_size() > 1:
            torch.distributed.barrier()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    def run(self, steps=None, spawn=False):
        steps = list(self.ORDER) if steps is None else [s for s in self.ORDER if s in set(steps)]
        if dist.rank() == 0:
            print(f"Start: {datetime.now()}", flush=True)
        if spawn and dist.world_size() == 1:
            import multiprocessing as mp
            ctx = mp.get_context("spawn")
            for key in steps:
                p = ctx.Process(target=_run_one, args=(dataclasses.asdict(self.o), key))
                p.start()
                p.join()
                if p.exitcode != 0:          # (the reference ignores the exit code and runs the next step on stale files)
                    raise RuntimeError(f"step {key} exited with code {p.exitcode}")
        else:
            for key in steps:
                self.run_step(key)
        if dist.rank() == 0:
            print(f"Finished: {datetime.now()}", flush=True)


def _run_one(option_dict, key):
    dist.init_from_env()
    Workflow(WorkflowOptions(**option_dict)).run_step(key)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--root", default="./", help="ROOT_DIR (holds Input_Masks/ and Input_Images/)")
    ap.add_argument("--steps", default=None, help="comma-separated subset of 0,1,2,3,4,5,6a,6b (default: all)")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE", help="override an option of StartProcess.py:14-43")
    ap.add_argument("--spawn", action="store_true", help="one fresh process per step, as the reference does")
    a = ap.parse_args(argv)
    opts = WorkflowOptions(ROOT_DIR=a.root)
    for kv in a.set:
        k, _, v = kv.partition("=")
        opts.set(k.strip(), v)
    dist.init_from_env()
    Workflow(opts).run(a.steps.split(",") if a.steps else None, spawn=a.spawn)


if __name__ == "__main__":
    main()
