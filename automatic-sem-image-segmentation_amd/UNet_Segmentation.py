"""MultiResUNet workflow with the reference's class / attribute names (Releases/Version 1.2.0/UNet_Segmentation.py),
executing on libsemseg_hip.so.

Reference surface mirrored here (file:line in the reference):
* ``ImageDataset`` / ``DataLoader`` / ``DataSet`` ... UNet_Segmentation.py:21-144
* ``UNet`` ctor + attribute defaults ............... UNet_Segmentation.py:147-205 (StartProcess.py:151-156 overrides)
* ``run_training`` / ``create_model`` .............. UNet_Segmentation.py:246-288, 363-396
* ``step_decay`` / ``linear_decay`` ................ UNet_Segmentation.py:233-244
* the Keras default ``train_step`` + metrics (loss / mae / acc) that ``model.fit`` runs (third party in the reference).
"""
import json
import os
import random
import threading
import time

import numpy as np
import torch

from . import HelperFunctions
from . import dist as D
from . import losses
from .engine import Act, Tape
from .nets import MultiResUNet
from .optim import Adam


# augmentation id -> axes np.flip reverses (UNet_Segmentation.py:93-98: none, left-right, up-down, both)
_FLIP_AXES = {0: None, 1: (1,), 2: (0,), 3: (0, 1)}


class ImageDataset:
    """The image / mask file pairs of one subset: 80/20 split of the shuffled directory listing (``random.Random(1234)``), every
    image under four flip ids (UNet_Segmentation.py:21-101).  ``image_ids`` / ``image_info`` keep the reference's names and record
    layout; decoding goes through one ``_decode`` per file kind."""

    def __init__(self, image_dir, mask_dir, contrast_optimization_range=(0.5, 99.5), use_brightness_and_contrast_augmentation=False):
        self.image_dir, self.mask_dir = image_dir, mask_dir
        self.contrast_optimization_range = contrast_optimization_range
        self.use_brightness_and_contrast_augmentation = use_brightness_and_contrast_augmentation
        self.type = ''
        self.image_ids, self.image_info = [], {}
        # decoded tiles by (path, kind): every file is read under four flip ids per epoch, for UNET_EPOCHS epochs, and decoding one
        # (PIL + two percentiles) costs ~3 ms -- 30 ms per batch of 5 against a ~12 ms train step.  Bounded (SS_LOADER_CACHE_MB, 0 = off);
        # what does not fit is decoded every time, as in the reference.  Masks are kept as uint8 (they are 0 / 1 after the threshold).
        self._cache, self._cache_bytes, self._cache_lock = {}, 0, threading.Lock()
        self.cache_limit_bytes = int(float(os.environ.get("SS_LOADER_CACHE_MB", "2048")) * 2 ** 20)

    def add_image(self, image_id, path, mask, augmentation):
        self.image_ids.append(image_id)
        self.image_info[image_id] = dict(id=image_id, image_path=path, mask_path=mask, augmentation=augmentation)

    def initialize_images(self, subset, train_val_split=0.8, seed=1234):
        if subset not in ("train", "val"):
            raise AssertionError(subset)
        self.type = subset
        files = HelperFunctions.get_image_file_paths_from_directory(self.image_dir)
        random.Random(seed).shuffle(files)
        cut = int(train_val_split * len(files))
        for i, path in enumerate(files[:cut] if subset == "train" else files[cut:]):
            mask = path.replace(self.image_dir, self.mask_dir)
            for flip in sorted(_FLIP_AXES):
                self.add_image(f"{i:05d}_augmentation_{flip}", path, mask, flip)

    def _decode(self, info, is_mask):
        load = HelperFunctions.load_and_preprocess_images
        if is_mask:
            return load(info['mask_path'], normalization_range=(0, 1), threshold_value=0.5)[0]
        if self.type == 'train' and self.use_brightness_and_contrast_augmentation:
            # three draws from the module-level RNG, in the reference's order: contrast window start, then the two range offsets
            lo_pct = random.random() * 2
            below, above = random.random(), random.random()
            image = load(info['image_path'], normalization_range=(0 - below, 1 + above), contrast_optimization_range=(lo_pct, lo_pct + 98))[0]
            image -= np.min(image)
            image /= np.max(image)
            return image
        return load(info['image_path'], normalization_range=(0, 1), contrast_optimization_range=self.contrast_optimization_range)[0]

    def _tile(self, info, is_mask):
        if self.cache_limit_bytes <= 0 or (not is_mask and self.type == 'train' and self.use_brightness_and_contrast_augmentation):
            return self._decode(info, is_mask)          # off, or a random contrast window per read
        path = info['mask_path' if is_mask else 'image_path']
        try:          # the file's identity and the decode parameters are part of the key: a rewritten file or another contrast window re-decodes
            stt = os.stat(path)
            stamp = (stt.st_mtime_ns, stt.st_size)
        except OSError:
            stamp = None
        key = (path, is_mask, stamp, None if is_mask else tuple(self.contrast_optimization_range or ()))
        tile = self._cache.get(key)
        if tile is None:
            tile = self._decode(info, is_mask)
            keep = tile.astype(np.uint8) if is_mask and tile.min() >= 0 and tile.max() <= 1 and np.all(tile == np.rint(tile)) else tile
            keep.setflags(write=False)
            with self._cache_lock:          # the prefetching threads may have decoded the same tile twice: the first copy stays
                if key not in self._cache and self._cache_bytes + keep.nbytes <= self.cache_limit_bytes:
                    self._cache[key] = keep
                    self._cache_bytes += keep.nbytes
        return tile

    def load_from_file(self, image_ids, is_mask):
        ids = [image_ids] if isinstance(image_ids, str) else image_ids
        out = []
        for image_id in ids:
            info = self.image_info[image_id]
            tile, axes = self._tile(info, is_mask), _FLIP_AXES[info['augmentation']]
            out.append(tile if axes is None else np.flip(tile, axis=axes))
        return np.asarray(out, dtype='float32')


class _Feeder:
    """What ``fit`` iterates (a ``keras.utils.Sequence`` in the reference): batches ``fetch(keys[i*N:(i+1)*N])`` over a key order that
    ``reorder`` permutes between epochs.  ``keep_partial`` is the length policy: ceil keeps the short last batch, floor drops it."""

    def __init__(self, keys, fetch, reorder, batch_size, shuffle, keep_partial):
        self._keys, self._fetch, self._reorder = keys, fetch, reorder
        self.batch_size, self.shuffle, self._keep_partial = batch_size, shuffle, keep_partial

    def __len__(self):
        n, b = len(self._keys), self.batch_size
        return -(-n // b) if self._keep_partial else n // b

    def __getitem__(self, idx):
        return self._fetch(self._keys[idx * self.batch_size:(idx + 1) * self.batch_size])

    def on_epoch_end(self):
        if self.shuffle:
            self._reorder(self._keys)


class DataLoader(_Feeder):
    """Batches decoded from disk on demand, partial last batch kept; ids reshuffled with ``np.random`` (UNet_Segmentation.py:104-121)."""

    def __init__(self, dataset, batch_size=1, shuffle=True, **kwargs):
        self.dataset = dataset
        super().__init__(list(dataset.image_ids), lambda ids: (dataset.load_from_file(ids, is_mask=False), dataset.load_from_file(ids, is_mask=True)),
                         np.random.shuffle, batch_size, shuffle, keep_partial=True)

    @property
    def all_image_ids(self):
        return self._keys


class DataSet(_Feeder):
    """Arrays held in memory, short last batch dropped; (x, y) pairs reshuffled with the ``random`` module (UNet_Segmentation.py:124-144)
    -- as one index permutation: ``random.shuffle`` moves positions, not values, so permuting ``range(n)`` draws the same numbers and
    lands every pair where shuffling the zipped list would."""

    def __init__(self, x, y, batch_size=1, shuffle=True, **kwargs):
        self.x, self.y = x, y
        super().__init__(range(len(x)), lambda sl: (self.x[sl.start:sl.stop], self.y[sl.start:sl.stop]), self._permute, batch_size, shuffle,
                         keep_partial=False)

    def _permute(self, _keys):
        order = list(range(len(self.x)))
        random.shuffle(order)
        self.x = np.asarray(self.x, dtype='float32')[order]
        self.y = np.asarray(self.y, dtype='float32')[order]


_NAN_TRAP = [os.environ.get("SS_NAN_TRAP", "0") == "1", 0]
_SYNC_ONLY = os.environ.get("SS_NAN_TRAP", "0") == "2"


class UNetModel:
    """What ``keras.models.Model(input, multi_res_unet).compile(loss=weighted_bce, optimizer=Adam, metrics=['mae','acc'])``
    provides to the workflow: ``train_step`` / ``test_step`` / ``predict`` (UNet_Segmentation.py:386-396)."""

    def __init__(self, net, weighting, optimizer):
        self.net, self.weighting, self.optimizer = net, float(weighting), optimizer
        self.device = net.device
        self._out3 = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.act_dtype = net.act_dtype           # float32, or bfloat16 / float16 mixed-precision activation storage
        self.loss_scale = (float(os.environ.get("SS_F16_LOSS_SCALE", "1024")) if self.act_dtype == torch.float16 else 1.0)
        # SS_UNET_WGRAD_STREAM=0: weight gradients on the chain's stream; "force": on the side stream even when several ranks share a GPU
        # (tests/test_dp_gpu.py: the bucket hooks must order themselves behind BOTH streams)
        self.wgrad_side_stream = {"0": False, "force": "force"}.get(os.environ.get("SS_UNET_WGRAD_STREAM", "1"), True)
        # measured: 36.5 ms per step without, 37.1 - 37.3 with (the refresh of ~100 small layers competes with the chain's first, small
        # layers): opt-in here, default in the CycleGAN step
        self.refresh_side_stream = os.environ.get("SS_UNET_REFRESH_STREAM", "0") == "1"
        # the same refresh as ONE recorded plan in front of the chain on the chain's own stream (no event, no second stream)
        self.refresh_inline = os.environ.get("SS_UNET_REFRESH_INLINE", "0") == "1"
        # sync_metrics = False: train_step returns {} and leaves the three scalars on the device (no device->host read), so the caller
        # can issue it on a stream of its own beside other work; side_stream_index = which of engine.side_streams the weight gradients take
        self.sync_metrics = True
        self.side_stream_index = 0
        # the four ResPaths on streams of their own (engine.Branch; nets.MultiResUNet.forward): SS_UNET_BRANCHES=0 runs them inline.
        # Single-process only: under data parallelism the SyncBN / gradient collectives would be issued from several streams.
        self.branch_streams = os.environ.get("SS_UNET_BRANCHES", "1") != "0"
        # which of engine.side_streams: [weight gradients, ResPath 1..4] (streams created one after the other land on the HIP runtime's
        # hardware queues round-robin, so the indices decide which chains share a queue); SS_UNET_STREAMS="w,b1,b2,b3,b4" overrides
        env = os.environ.get("SS_UNET_STREAMS")
        self.stream_indices = [int(v) for v in env.split(",")] if env else None
        # OPT-IN (graph = True / SS_UNET_GRAPH=1): the step -- forward, loss, backward on its side streams, Adam -- captured ONCE per input
        # shape into a hipGraph (torch.cuda.graph) and replayed: same kernels, same order per stream, same bits
        # (tests/test_nets_gpu.py); what changes per step enters through device memory (the input tiles, Adam's alpha).  Not the
        # default: at per-GPU batch 1 (the 8-GPU share of BASELINE config 4: one 512 x 512 tile, ~1100 launches) the step is paced by
        # the per-dispatch cost on BOTH sides -- eager: 12.8 ms of host issue, 11.1 ms wall; replayed: hipGraphLaunch of the
        # 1100-node graph still costs 9.2 ms of host time (ROCm 7.2: ~8 us per node) and the wall time is 11.0 ms.  256 x 256 tiles:
        # 11.9 -> 9.1 ms (tools/unet_graph_probe.py, profiles/r05_unet_hipgraph_probe.txt).  Fewer launches help, a graph does not.
        self.graph = os.environ.get("SS_UNET_GRAPH", "0") == "1"
        self._graphs = {}          # (n, h, w) -> eager step count (int, warm-up) or the captured state (dict)

    def _to_act(self, t):
        from .engine import convert
        if isinstance(t, Act):          # already on the device (as CycleGanModel accepts them)
            return convert(t, self.act_dtype)
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        return convert(Act(t.to(self.device, dtype=torch.float32).contiguous(), requires_grad=False), self.act_dtype)

    GRAPH_WARMUP_STEPS = 2

    def train_step(self, batch):
        """fwd(training=True) -> class-weighted BCE -> backward -> Adam.  Returns {'loss','mae','acc'} of this batch."""
        if self._graph_wanted(batch):
            return self._train_step_graph(batch)
        x, y = (self._to_act(t) for t in batch)
        self._issue_step(x, y)
        return self._read_metrics()

    def _issue_step(self, x, y, alpha_dev=None):
        """Enqueue one optimisation step on the current stream (+ its side streams); nothing here waits for the device."""
        world = D.world_size()
        tape = Tape()
        if self.wgrad_side_stream and x.device.type == "cuda" and (self.wgrad_side_stream == "force" or not D.ranks_share_device()):
            # the weight gradients of the tile-kernel layers (matrix cores / LDS) beside the BatchNorm backward passes of the chain
            # (HBM): they are off the dependency chain (engine.Tape.wgrad_stream)
            from .engine import side_streams
            k = self.side_stream_index
            idx = self.stream_indices if self.stream_indices is not None else [k, k + 1, k + 2, k + 1, k + 2]      # measured: 33.8 ms against 34.8 - 35.2 on five streams
            streams = side_streams(x.device, max(idx) + 1)
            tape.wgrad_stream = streams[idx[0]]
            if self.branch_streams and world == 1:
                tape.branch_streams = [streams[i] for i in idx[1:]]
            if self.refresh_side_stream:          # the layers' weight-derived operands beside the first layers, not inside the chain
                self.net.arena.refresh_derived(side_streams(x.device, 7)[6])
        if self.refresh_inline and not self.refresh_side_stream and x.device.type == "cuda":
            self.net.arena.refresh_derived()          # the recorded plan (engine.ParamArena._refresh_batched) in front of the chain, on its stream
        p = self.net(x, True, tape)
        losses.weighted_bce(y, p, self.weighting, self.loss_scale, self._out3)
        self.net.zero_grad()
        D.begin_backward([self.net])
        tape.backward()
        D.all_reduce_grads([self.net])
        if _SYNC_ONLY:            # diagnostics (SS_NAN_TRAP=2): a device-wide synchronisation between backward and the optimizer step, nothing else
            torch.cuda.synchronize()
        if _NAN_TRAP[0]:          # diagnostics (SS_NAN_TRAP=1): the first step whose output or gradients are not finite, and where
            torch.cuda.synchronize()
            a = self.net.arena
            bad = [(n, int((~torch.isfinite(a.gviews[n])).sum())) for n in a.gviews if not bool(torch.isfinite(a.gviews[n]).all())]
            pf = bool(torch.isfinite(p.dense().float()).all())
            _NAN_TRAP[1] += 1
            if bad or not pf:
                print(f"SS_NAN_TRAP: step {_NAN_TRAP[1]}: output finite: {pf}; {len(bad)} of {len(a.gviews)} gradients not finite; first: {bad[:10]}; last: {bad[-4:]}",
                      flush=True)
                _NAN_TRAP[0] = False
        self.optimizer.apply(self.net, 1.0 / (world * self.loss_scale), alpha_dev=alpha_dev)

    def _read_metrics(self):
        if not self.sync_metrics:
            return {}
        # THIS rank's values (its shard of the batch): the cross-rank mean is taken once per logging interval (global_metrics), not per step
        s = self._out3.cpu().numpy().astype(np.float64)
        return {"loss": float(s[0]), "mae": float(s[1]), "acc": float(s[2])}

    @staticmethod
    def global_metrics(m):
        """Mean over the ranks of a dict of per-rank means (equal shards) -- one small all-reduce; single process: unchanged."""
        keys = sorted(m)
        v = D.mean_scalars(np.array([m[k] for k in keys], dtype=np.float64))
        return {k: float(v[i]) for i, k in enumerate(keys)}

    # ---- the step as a replayed hipGraph -------------------------------------------------------------------------------------------
    @staticmethod
    def _shape_of(t):
        return tuple(int(v) for v in (t.t.shape if isinstance(t, Act) else t.shape))

    def _graph_wanted(self, batch):
        if not self.graph or self.device.type != "cuda" or D.world_size() != 1 or D.is_dist():
            return False
        from . import layers as LY
        if LY.SYNC_BN is not None:
            return False
        shp = self._shape_of(batch[0])
        if len(shp) != 4 or self._shape_of(batch[1]) != shp or shp[3] != 1:          # the static buffers are one-channel: others stay eager
            return False
        return True

    @staticmethod
    def _as_f32(t, like):
        if isinstance(t, Act):
            t = t.dense() if (t.c0 or t.c != t.cs) else t.t
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        return t.to(device=like.device, dtype=torch.float32)

    def _train_step_graph(self, batch):
        from .engine import capture_scope
        key = self._shape_of(batch[0])[:3]
        st = self._graphs.get(key, 0)
        if isinstance(st, int):
            if st < self.GRAPH_WARMUP_STEPS:          # eager first: scratch sizes, weight caches and one-time kernel attributes settle
                self._graphs[key] = st + 1
                x, y = (self._to_act(t) for t in batch)
                self._issue_step(x, y)
                return self._read_metrics()
            n, h, w = key
            st = dict(x=torch.empty((n, h, w, 1), dtype=torch.float32, device=self.device),
                      y=torch.empty((n, h, w, 1), dtype=torch.float32, device=self.device),
                      alpha=torch.zeros(1, dtype=torch.float32, device=self.device), graph=torch.cuda.CUDAGraph())
            st["x"].copy_(self._as_f32(batch[0], st["x"]))
            st["y"].copy_(self._as_f32(batch[1], st["y"]))
            iters = self.optimizer.iterations
            try:
                with capture_scope() as scope, torch.cuda.graph(st["graph"]):
                    from .engine import convert
                    xa = convert(Act(st["x"], requires_grad=False), self.act_dtype)
                    ya = convert(Act(st["y"], requires_grad=False), self.act_dtype)
                    self._issue_step(xa, ya, alpha_dev=st["alpha"])
            except Exception as e:          # noqa: BLE001  -- a capture this runtime refuses: say so once, stay eager (same results)
                self.optimizer.iterations = iters
                self.graph = False
                self.net.arena.touch()
                D.warn_once(f"UNetModel: hipGraph capture of the train step failed ({type(e).__name__}: {e}); running eagerly")
                return self.train_step(batch)
            st["kept"] = scope.kept
            self._graphs[key] = st
            # capture launches nothing: the weights are untouched, this step runs as the first replay below
        else:
            st["x"].copy_(self._as_f32(batch[0], st["x"]))
            st["y"].copy_(self._as_f32(batch[1], st["y"]))
        st["alpha"].fill_(self.optimizer.next_alpha())
        st["graph"].replay()
        self.net.arena.touch()          # the replayed Adam changed the weights: derived operands kept by the layers are stale for eager calls
        return self._read_metrics()

    def test_step(self, batch):
        x, y = (self._to_act(t) for t in batch)
        p = self.net(x, False)
        losses.weighted_bce(y, p, self.weighting, 1.0, self._out3, want_grad=False)
        s = self._out3.cpu().numpy().astype(np.float64)
        return {"loss": float(s[0]), "mae": float(s[1]), "acc": float(s[2])}

    def predict(self, x, training=False):
        return self.net(self._to_act(x), training).dense().float()          # fp32 whatever the activation storage

    __call__ = predict

    def get_weights(self):
        return self.net.get_weights()

    def set_weights(self, w):
        self.net.set_weights(w)

    def save(self, path):
        """``model.save('…/model.keras')`` (UNet_Segmentation.py:262-264,287): Keras-3 archive (keras_io.py) with the network's layers
        and the Adam state; ``.npz`` paths give the plain-numpy form."""
        cfg = dict(filters=self.net.filters, weighting=self.weighting, learning_rate=float(self.optimizer.learning_rate),
                   output_channels=self.net.output_channels)
        if path.endswith(".npz"):
            arrays = {name: w for name, w in zip(self.net.variable_names, self.net.get_weights())}
            arrays["__config__"] = np.array(json.dumps(cfg))
            np.savez(path, **arrays)
            return
        from . import keras_io as K
        arrays = K.net_arrays(self.net, "")
        arrays.update(K.optimizer_arrays(self.optimizer, self.net, "optimizer/"))
        K.write_archive(path, arrays, "MultiResUNet", cfg)

    @classmethod
    def load(cls, path, device=None):
        """Counterpart of ``keras.models.load_model(.../model.keras, custom_objects={'weighted_bce': ...})`` (UNet_Segmentation.py:303)."""
        device = device if device is not None else D.local_device()
        if path.endswith(".npz"):
            z = np.load(path)
            cfg = json.loads(str(z["__config__"]))
            net = MultiResUNet(conv_filters=cfg["filters"], device=device, output_channels=cfg.get("output_channels", 1))
            net.set_weights([z[name] for name in net.variable_names])
            return cls(net, cfg.get("weighting", 1.0), Adam(cfg.get("learning_rate", 1e-3)))
        from . import keras_io as K
        _, cfg, arrays = K.read_archive(path)
        net = MultiResUNet(conv_filters=cfg["filters"], device=device, output_channels=cfg.get("output_channels", 1))
        K.load_net_arrays(net, "", arrays, K.NameCounters())
        model = cls(net, cfg.get("weighting", 1.0), Adam(cfg.get("learning_rate", 1e-3)))
        K.load_optimizer_arrays(model.optimizer, net, "optimizer/", arrays)
        return model


class UNet:
    def __init__(self, root_dir, image_dir, mask_dir, allow_memory_growth=True, use_gpus_no=(0,)):
        self.root_dir = os.path.join(root_dir, '3_UNet')
        self.model_dir = os.path.join(self.root_dir, "Models")
        self.image_dir, self.mask_dir = image_dir, mask_dir
        self.use_dataloader = False
        self.contrast_optimization_range = (1, 99)
        self.prefix = time.strftime('%Y-%m-%d_%H-%M-%S', time.localtime())
        self.batch_size = 1
        self.epochs = 100
        self.learning_rate = 0.001
        self.loss_function = 'binary_crossentropy'
        self.lr_decay = 'STEP_DECAY'
        self.image_shape = (384, 384, 1)
        self.filters = 16
        self.output_channels = 1
        self.allow_memory_growth = allow_memory_growth
        self.use_gpus_no = use_gpus_no
        self.activation_storage = os.environ.get("SS_ACT_DTYPE", "f32")     # not in the reference; see CycleGAN.CycleGAN
        self.dataset_train = self.dataset_val = None
        self.training_data = self.validation_data = None
        self.model = None
        self.device = D.local_device()
        self.seed = 0
        self.sync_batch_norm = True   # data parallel: whole-(global)-batch BatchNorm statistics, as on a single device

    def load_images(self, subset):
        assert subset in ['train', 'val']
        ds = self.dataset_train if subset == "train" else self.dataset_val
        if self.use_dataloader:
            return DataLoader(ds, self.batch_size)
        x = ds.load_from_file(ds.image_ids, is_mask=False)
        y = ds.load_from_file(ds.image_ids, is_mask=True)
        return DataSet(x, y, self.batch_size)

    def step_decay(self, epoch, current_lr, drop=0.5, epochs_drop=10):
        return current_lr * drop if (epoch + 1) % epochs_drop == 0 else current_lr

    def linear_decay(self, epoch, current_lr):
        return self.learning_rate * (1 - (epoch / float(self.epochs))) ** 1

    def class_weighting(self):
        """#zeros / #ones over all training masks (UNet_Segmentation.py:364-376)."""
        if self.use_dataloader:
            zeros = ones = 0
            tmp = None
            for image_id in self.dataset_train.image_ids:
                tmp = np.array(self.dataset_train.load_from_file(image_id, is_mask=True))
                zeros += np.count_nonzero(tmp == 0)
                ones += np.count_nonzero(tmp)
            self.image_shape = tmp.shape[1:3]
            return zeros / ones
        y = self.training_data.y
        self.image_shape = y.shape[1:3]
        return np.count_nonzero(y == 0) / np.count_nonzero(y)

    def create_model(self, weighting=None):
        if weighting is None:
            weighting = self.class_weighting()
        # output_channels > 1: Conv2D + softmax head and one-hot targets (UNet_Segmentation.py:387, 558-560); the reference's own
        # feeders only ever produce one-channel masks, and its inference path rebuilds a one-channel model (UNet_Segmentation.py:317)
        net = MultiResUNet(conv_filters=self.filters, device=self.device, seed=self.seed, output_channels=self.output_channels,
                           act_dtype=self.activation_storage)
        D.broadcast_params([net])
        D.enable_overlap([net])
        if D.world_size() > 1 and self.sync_batch_norm:
            D.enable_sync_bn(True)
        wd = self.lr_decay if isinstance(self.lr_decay, float) else 0.0
        return UNetModel(net, weighting, Adam(learning_rate=self.learning_rate, weight_decay=wd))

    def run_inference(self, files, output_directory, model=None, tile_images=False, threshold=-1, watershed_lines=True,
                      min_distance=9, min_overlap=2, manage_overlap_mode=2, use_gpu=False):
        """UNet_Segmentation.py:290-351: probabilities (``*_raw.tif``, float32) and the thresholded label map per image.
        Runs on the MI355X in inference mode (BatchNorm moving statistics).  ``watershed_lines=True`` (the reference default)
        splits touching particles on the CPU (HelperFunctions.segment -> libsemseg_post.so), as the reference does."""
        from PIL import Image
        if model is None and self.model is None:
            latest = sorted(os.listdir(self.model_dir))[-1]
            self.model = UNetModel.load(os.path.join(self.model_dir, latest, 'model.keras'), self.device)
        elif isinstance(model, str):
            self.model = UNetModel.load(model, self.device)
        elif model is not None:
            self.model = model
        input_files = HelperFunctions.load_and_preprocess_images(files, normalization_range=(0, 1),
                                                                 contrast_optimization_range=self.contrast_optimization_range)
        file_names = HelperFunctions.get_image_file_paths_from_directory(files) if isinstance(files, str) and os.path.isdir(files) \
            else ([files] if isinstance(files, str) else list(files))
        os.makedirs(output_directory, exist_ok=True)
        for i in range(input_files.shape[0]):
            input_file = input_files[i]
            if tile_images:
                tiles = np.array(HelperFunctions.tile_image(input_file, self.image_shape[0], self.image_shape[1], min_overlap=min_overlap))
                pred = np.array([self.model.predict(t[None])[0].cpu().numpy() for t in tiles])
                img = HelperFunctions.stitch_image(pred, input_file.shape[1], input_file.shape[0], min_overlap=min_overlap,
                                                   manage_overlap_mode=manage_overlap_mode)
            else:
                img = self.model.predict(np.ascontiguousarray(input_file[None]))[0].cpu().numpy().copy()
            img = img[:, :, 0]
            base = os.path.split(file_names[i])[-1]
            Image.fromarray(img).save(os.path.join(output_directory, base.replace(os.path.splitext(base)[-1], '_raw.tif')))
            img -= np.min(img)
            img /= np.max(img)
            img *= 255
            img = img.astype(np.uint8)
            img = HelperFunctions.segment(image=img, threshold=threshold, watershed_lines=watershed_lines, min_distance=min_distance,
                                          use_four_connectivity=True)
            Image.fromarray(img).save(os.path.join(output_directory, base))

    def run_training(self):
        """Equivalent of ``model.fit(training_data, epochs, callbacks, validation_data)`` (UNet_Segmentation.py:246-288)."""
        os.makedirs(os.path.join(self.model_dir, self.prefix), exist_ok=True)
        from . import keras_io
        keras_io.warn_if_no_hdf5('UNet.run_training')
        self.dataset_train = ImageDataset(self.image_dir, self.mask_dir, self.contrast_optimization_range)
        self.dataset_val = ImageDataset(self.image_dir, self.mask_dir, self.contrast_optimization_range)
        self.dataset_train.initialize_images('train')
        self.dataset_val.initialize_images('val')
        self.training_data = self.load_images('train')
        self.validation_data = self.load_images('val')
        self.model = self.create_model()
        log_path = os.path.join(self.model_dir, self.prefix, 'training_log.csv')
        best = float('inf')
        rank, world = D.rank(), D.world_size()
        D.check_batch_divisible(self.batch_size, world, 'UNet.batch_size')
        for epoch in range(self.epochs):
            if self.lr_decay == 'STEP_DECAY':
                self.model.optimizer.learning_rate = self.step_decay(epoch, self.model.optimizer.learning_rate)
            elif self.lr_decay == 'LINEAR_DECAY':
                self.model.optimizer.learning_rate = self.linear_decay(epoch, self.model.optimizer.learning_rate)
            tot = {"loss": 0.0, "mae": 0.0, "acc": 0.0}
            seen = 0
            order = list(range(len(self.training_data)))
            np.random.shuffle(order)
            # batches decoded ahead of the train steps (HelperFunctions.prefetch); not when the loader draws random numbers per batch
            ahead = HelperFunctions.PREFETCH_DEPTH if (self.use_dataloader and not getattr(self.dataset_train, "use_brightness_and_contrast_augmentation", False)) else 0
            for x, y in HelperFunctions.prefetch(self.training_data.__getitem__, order, depth=ahead):
                # ragged last batch of the on-demand loader (ceil length, UNet_Segmentation.py:111-112): every rank trims it to the
                # same multiple of the world size, so the collectives stay matched; a batch smaller than the world is skipped
                per = len(x) // world
                if per == 0:
                    D.warn_once('a partial batch smaller than the number of ranks was skipped (data parallel)')
                    continue
                if per * world != len(x):
                    D.warn_once('partial last batch trimmed to a multiple of the number of ranks (data parallel)')
                    x, y = x[:per * world], y[:per * world]
                m = self.model.train_step((x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]))
                for k in tot:
                    tot[k] += m[k] * len(x)     # Keras weights the running means by batch size
                seen += len(x)
            self.training_data.on_epoch_end()
            logs = {k: v / max(seen, 1) for k, v in tot.items()}
            if world > 1:
                logs = self.model.global_metrics(logs)          # the epoch's one metrics exchange (the steps return rank-local values)
            vt = {"loss": 0.0, "mae": 0.0, "acc": 0.0}
            vseen = 0
            for x, y in HelperFunctions.prefetch(self.validation_data.__getitem__, range(len(self.validation_data)), depth=ahead):
                m = self.model.test_step((x, y))
                for k in vt:
                    vt[k] += m[k] * len(x)
                vseen += len(x)
            logs.update({"val_" + k: v / max(vseen, 1) for k, v in vt.items()})
            if rank == 0:
                new = not os.path.exists(log_path)
                with open(log_path, 'a') as f:
                    if new:
                        f.write(';'.join(['epoch'] + sorted(logs)) + '\n')
                    f.write(';'.join([str(epoch)] + [repr(logs[k]) for k in sorted(logs)]) + '\n')
                if logs["loss"] < best:
                    best = logs["loss"]
                    self.model.save(os.path.join(self.model_dir, self.prefix, "Checkpoint_Lowest_Loss.keras"))
        if rank == 0:
            self.model.save(os.path.join(self.model_dir, self.prefix, 'model.keras'))
        return self.model
