"""Attack drivers: the host-side mirror of the reference's four attack scripts around the HIP kernels.

    PgdAttack     attack/DSGN/pgd_attack.py:229-374 and attack/Stereo-RCNN/pgd_attack.py:105-243
    PatchTrainer  attack/DSGN/patch_attack.py:278-443 and attack/Stereo-RCNN/patch_attack.py:99-293

The detector is the caller's: anything with ``loss_and_grad(x, extra) -> (loss, grad)`` where ``x`` is
the stacked stereo batch [2B,3,H,W] (left eyes first, then right eyes) and ``grad`` = d loss / d x.
Keeping both eyes of all B pairs in ONE buffer is what lets a PGD step be a single kernel launch.

Differences from the reference, all deliberate and listed in DESIGN.md: any batch size B (the
reference's in-place denormalize only works for B = 1, quirk Q1); one clean buffer instead of two
identical ones (Q2); explicit clone of the Stereo R-CNN clean image (Q6); a seedable patch-position
stream (Q9); device-side loss accumulation (Q17); PNG encoding off the critical path.
"""
import itertools
import os

import torch

from . import patchgeom, pixelio
from .determinism import under_solvers
from .dist import Comm


def split_eyes(x):
    """views (imgL, imgR) of a stacked [2B,3,H,W] batch"""
    b = x.shape[0] // 2
    return x[:b], x[b:]


def _owned(loader, comm):
    """(global batch index, batch) of the batches this rank owns (i % world == rank).  Loaders with ``shard`` and
    ``__len__`` (data.SyntheticStereo, data.KittiFolder) never materialise the other ranks' batches."""
    if comm.world > 1 and hasattr(loader, "shard"):
        for k, batch in enumerate(loader.shard(comm.rank, comm.world)):
            yield comm.rank + k * comm.world, batch
        return
    for i, batch in enumerate(loader):
        if i % comm.world == comm.rank:
            yield i, batch


def _stack_on_device(batch, dev):
    """[2B,3,H,W] float32 on ``dev``, left eyes first; page-locked inputs (data.KittiFolder) are copied without blocking"""
    b = len(batch)
    if batch.imgL.device.type == "cpu" and torch.device(dev).type == "cuda" and batch.imgL.dtype == torch.float32:
        x = torch.empty((2 * b,) + tuple(batch.imgL.shape[1:]), dtype=torch.float32, device=dev)
        x[:b].copy_(batch.imgL, non_blocking=batch.imgL.is_pinned())
        x[b:].copy_(batch.imgR, non_blocking=batch.imgR.is_pinned())
        return x
    return torch.cat([batch.imgL, batch.imgR], dim=0).to(dev, dtype=torch.float32).contiguous()


def _default_ops():
    from . import ops       # raises if libadvengine.so is not built: there is no other compute path
    return ops


class StereoBatch:
    """What a loader hands to the drivers (the reference's BatchCollator dict, attack/DSGN/pgd_attack.py:103-126,
    reduced to what the loop itself touches).  imgL/imgR: float32 [B,3,H,W] in the detector's input space;
    names: file stem per pair; sizes: (w, h) of the original image per pair or None; extra: opaque, handed to
    the model adapter (calibration, targets, depth map ...).  A loader may instead hand over the raw 8-bit RGB pixels,
    uint8 [B,h,w,3] with ``pad_to`` = the network frame: the DSGN transform then runs on the device (ops.import_u8)."""

    def __init__(self, imgL, imgR, names, sizes=None, extra=None, pad_to=None):
        if imgL.dtype == torch.uint8:      # raw 8-bit pixels [B,h,w,3] for the device-side loader transform; pad_to = network (H, W)
            assert imgL.shape == imgR.shape and imgL.dim() == 4 and imgL.shape[3] == 3 and pad_to is not None
        else:
            assert imgL.shape == imgR.shape and imgL.dim() == 4 and imgL.shape[1] == 3
        self.imgL, self.imgR, self.names, self.sizes, self.extra, self.pad_to = imgL, imgR, list(names), sizes, extra, pad_to

    def __len__(self):
        return self.imgL.shape[0]


class PgdAttack:
    """Iterative L-inf PGD / FGSM (iters = 1) through a stereo detector.

    model_kind 'dsgn':  pixel space = ImageNet-normalised RGB, files ``dsgn_pgd_iters_{k}/image_{2,3}/{name}.png``
                        cropped to the original (w, h), 8-bit by truncation (attack/DSGN/pgd_attack.py:157-193).
    model_kind 'srcnn': pixel space = BGR minus PIXEL_MEANS, ``eps`` is the script argument and is scaled by
                        255 here (attack/Stereo-RCNN/pgd_attack.py:57), files ``stereo_rcnn_pgd_iters_{k}/...``
                        at network scale (quirk Q14).  Iterate 0 is written to the intended ``image_2/<name>``
                        path, not the reference's ``image_2<name>`` (quirk Q5).
    The loss is ASCENDED (untargeted), as in both scripts.
    """

    def __init__(self, model_kind, alpha, eps, iters, out_root=".", save=True, save_every=1, writer_workers=None,
                 ops=None, device=None, in_place=True, reference_on_gpu=False, graph=False, png_compress_level=1):
        if writer_workers is None:                   # PNG (zlib) encoding is the I/O wall: 42 files per pair at N = 20; zlib releases
            writer_workers = min(96, max(4, (os.cpu_count() or 8) // 2))     # the GIL, so the pool scales with the host's cores
        self.ops = ops if ops is not None else _default_ops()
        self.kind = model_kind
        if model_kind == "dsgn":
            # reference_on_gpu: reproduce a GPU run of the reference script bit for bit (torch's GPU kernels multiply by the
            # reciprocal where the CPU ones divide); the default reproduces its CPU run (arithmetic contract)
            self.space = self.ops.Space.dsgn(reference_on_gpu=True) if reference_on_gpu else self.ops.Space.dsgn()
            self.prefix, self.eps = "dsgn", float(eps)
        elif model_kind == "srcnn":
            self.space, self.prefix, self.eps = self.ops.Space.srcnn(), "stereo_rcnn", 255 * float(eps)
        else:
            raise ValueError("model_kind must be 'dsgn' or 'srcnn'")
        self.alpha, self.iters = float(alpha), int(iters)
        self.out_root, self.save, self.save_every = out_root, save, max(1, int(save_every))
        self.device, self.in_place = device, in_place
        # graph=True: ONE iteration - detector forward + loss + backward + the fused PGD step (+ its 8-bit export) - is captured in a
        # hipGraph (torch's stream capture) and replayed ``iters`` times: one launch per iteration instead of ~10^3.  Needs a detector
        # without host read-backs or data-dependent shapes (adapters.PsvStereoAdapter / DsgnShapedAdapter; not the proposal-based
        # Stereo R-CNN graphs); same bits as the eager loop (tests/test_gpu_drivers.py).
        self.graph = bool(graph)
        self.max_graphs = 4          # captures kept for reuse (graph=True): one per label signature, least recently used dropped
        self.writer = pixelio.PngWriter(writer_workers, bgr=(model_kind == "srcnn"), compress_level=png_compress_level) if save else None

    # -- file surface --------------------------------------------------------------------------
    def _wanted(self, k):
        return self.save and (k % self.save_every == 0 or k == self.iters)

    def _fan_out(self, k, batch):
        b = len(batch)
        names, sizes = list(batch.names), batch.sizes

        def fan_out(host):                           # host: [2B, rows, W, 3] uint8, private copy
            for eye in (0, 1):
                d = os.path.join(self.out_root, pixelio.iter_dir(self.prefix, k, eye))
                for i, name in enumerate(names):
                    stem = name if name.lower().endswith(".png") else name + ".png"
                    w, h = sizes[i] if sizes is not None else (None, None)
                    self.writer.put(os.path.join(d, stem), host[eye * b + i], crop_w=w, crop_h=h)
        return fan_out

    # -- one batch -----------------------------------------------------------------------------
    @under_solvers
    def run_batch(self, batch, adapter):
        """Attack B stereo pairs; returns the final stacked iterate [2B,3,H,W] (device).  Runs under determinism.solvers(); this package's
        adapters warm a new input shape up by themselves (determinism.deterministic): the iterates are a function of the inputs only."""
        ops, sp = self.ops, self.space
        dev = self.device if self.device is not None else batch.imgL.device
        if self.graph and getattr(self, "_graph_cache", None) is not None and torch.cuda.is_available():
            # a captured iteration is about to be replayed again: let whatever the caller did with the previous result (a clone, a loss
            # read-back) finish before this batch's eager preparation starts - third of the three waits that keep a reused capture of the
            # Stereo R-CNN-shaped step from faulting on this stack (see _run_graph); once per batch
            torch.cuda.synchronize(dev)
        imported = None
        if batch.imgL.dtype == torch.uint8:
            # 8-bit HWC pixels from the loader (data.KittiFolder(as_u8=True)): ToTensor + Normalize + zero padding run on the device,
            # and the clean image and its index come with them (ops.import_u8) - a quarter of the upload, no host conversion
            b2 = len(batch)
            u8 = torch.empty((2 * b2,) + tuple(batch.imgL.shape[1:]), dtype=torch.uint8, device=dev)    # no concatenation on the host:
            u8[:b2].copy_(batch.imgL, non_blocking=batch.imgL.is_pinned())                               # reading page-locked memory
            u8[b2:].copy_(batch.imgR, non_blocking=batch.imgR.is_pinned())                               # back on the CPU is slow
            valid = None if batch.sizes is None else [(s[1], s[0]) for s in batch.sizes] * 2
            # the loader's arithmetic is true divisions whichever way the steps re-normalise: import with the plain DSGN space
            # (adv_import_u8_f32 takes no AFFINE_RCP space), keep ``sp`` (possibly the reference-on-GPU space) for the steps
            imported = ops.import_u8(u8, ops.Space.dsgn() if sp.affine else sp, batch.pad_to, valid=valid)
            x = imported[0]
        else:
            x = _stack_on_device(batch, dev)
        n, _, h, w = x.shape
        rows = h if batch.sizes is None else max(s[1] for s in batch.sizes)
        cols = w if batch.sizes is None else max(s[0] for s in batch.sizes)
        exporter = pixelio.AsyncExporter(self.writer, lambda: ops.alloc_u8(n, rows, w, dev), dev) if self.save else None
        # clean image in pixel space: pgd_attack.py:297-298 (DSGN) / :122-123 (Stereo R-CNN)
        # (for 8-bit derived inputs the clean image is also kept as one byte per element: the N steps then read it
        #  as bytes; verified per image on the device, results identical - ops.denormalize_indexed.  The loader's zero
        #  padding beyond each image's own (h, w) is part of what is verified.)
        cidx = None
        want0 = self._wanted(0)                      # iterate 0 = the un-attacked pair, :279-294
        if imported is not None:
            _, clean, cidx = imported
            if want0:
                ops.export_u8(x, sp, (rows, cols), out=exporter.next_buffer())
        elif getattr(ops, "can_index_clean", lambda *_: False)(x, sp):
            valid = None if (batch.sizes is None or not sp.affine) else [(s[1], s[0]) for s in batch.sizes] * 2     # both eyes
            clean, cidx = ops.denormalize_indexed(x, sp, valid=valid, u8_out=exporter.next_buffer() if want0 else None,
                                                  crop=(rows, cols) if want0 else None)
        else:
            clean = ops.denormalize(x, sp) if sp.affine else x.clone()
            if want0:
                ops.export_u8(x, sp, (rows, cols), out=exporter.next_buffer())
        if want0:
            exporter.submit(self._fan_out(0, batch))
        self.last_clean_index = cidx
        losses = []
        # In place by default: every step kernel (also the line-aligned one for Stereo R-CNN's planes) reads only the
        # elements it writes.  Two alternating buffers measured the same within 1 % (profiles/r02_k1_idx_tuning.md);
        # ``in_place=False`` keeps the previous iterate intact for callers that want it.
        pingpong = not self.in_place
        spare = torch.empty_like(x) if pingpong else None
        if self.graph and self.iters > 0:
            x = self._run_graph(x, clean, cidx, adapter, batch, exporter, rows, cols, losses)
            if exporter is not None:
                exporter.close()
            self.last_losses = losses
            return x
        for k in range(self.iters):
            loss, grad = adapter.loss_and_grad(x, batch.extra)           # detector fwd + loss + bwd (:305-336)
            losses.append(loss)
            want = self._wanted(k + 1)
            nxt = spare if pingpong else x
            ops.pgd_step(x, grad.contiguous(), clean, sp, self.alpha, self.eps, out=nxt,
                         u8_out=exporter.next_buffer() if want else None,
                         crop=(rows, cols) if want else None,            # :339-354 (+ :357-374 export)
                         **({"clean_index": cidx} if cidx is not None else {}))
            if pingpong:
                x, spare = nxt, x
            if want:
                exporter.submit(self._fan_out(k + 1, batch))
        if exporter is not None:
            exporter.close()
        self.last_losses = losses
        return x

    def _run_graph(self, x, clean, cidx, adapter, batch, exporter, rows, cols, losses):
        """One iteration captured on STATIC buffers and replayed ``iters`` times.  The captured graph is kept and reused for the next
        batch when everything baked into it is unchanged - the adapter, the iterate's shape, the kind of clean image (indexed or
        float) and the batch's ``extra`` OBJECT (label-derived index lists of data-dependent length are part of the captured kernels'
        arguments: another label set needs another capture) - the new batch's iterate / clean image / index are then copied into the
        static buffers.  Otherwise: 2 eager warm-up iterations + capture + instantiate, once per batch."""
        ops, sp = self.ops, self.space
        if not self.in_place:
            raise ValueError("graph=True replays one captured iteration on one buffer: it needs in_place=True")
        if not getattr(adapter, "graph_safe", False):
            raise ValueError("%s does not declare graph_safe: a detector step with host read-backs or data-dependent shapes (upstream models; the "
                             "Stereo R-CNN-shaped surrogates unless their static forward is opted in: allow_graph_capture) cannot be captured in a "
                             "hipGraph" % type(adapter).__name__)
        any_export = exporter is not None and any(self._wanted(k + 1) for k in range(self.iters))
        # everything baked into the captured kernels' arguments besides the buffers: shapes, the kind of clean image, the export geometry -
        # and the step's own constants (alpha, eps, the pixel space); the adapter OBJECT is held by the capture and compared with ``is``
        key = (tuple(x.shape), x.device, cidx is not None, None if cidx is None else isinstance(cidx.valid, torch.Tensor), any_export, rows, cols,
               float(self.alpha), float(self.eps), getattr(sp, "name", None))
        # another batch's labels fit a capture when they are the same OBJECT - or when the adapter can say so: ``graph_extra_signature``
        # (everything of the labels that is a host constant of the capture: counts, sizes, shapes) equal, and the label TENSORS copied into
        # the captured ones (``graph_copy_extra``).  Up to ``max_graphs`` captures are kept (each holds its private memory pool).
        sig_fn = getattr(adapter, "graph_extra_signature", None)
        sig = sig_fn(batch.extra) if sig_fn is not None else None
        pool = self.__dict__.setdefault("_graph_caches", [])
        held = None
        for h in pool:
            if h["adapter"] is adapter and h["key"] == key and (h["extra"] is batch.extra or (sig is not None and h["sig"] == sig)) and \
                    (cidx is None or isinstance(cidx.valid, torch.Tensor) or tuple(cidx.valid) == tuple(h["cidx"].valid)):
                held = h
                break
        if held is not None:
            pool.remove(held)
            pool.append(held)                                # most recently used last
            self._graph_cache = held
            xs = held["x"]
            # the static buffers are about to be rewritten by eager copies: nothing of the previous batch's replays may still be in flight
            # (stream order should see to that; without this wait a reused capture of the Stereo R-CNN-shaped step faulted on replay - a race
            # that a device-wide wait before the copies removes, found by tracing, root cause below the runtime's surface; once per batch)
            torch.cuda.synchronize(x.device)
            xs.copy_(x)
            held["clean"].copy_(clean)
            if held["extra"] is not batch.extra:
                adapter.graph_copy_extra(held["extra"], batch.extra)
            if cidx is not None:
                for name in ("index", "ok", "lut"):
                    getattr(held["cidx"], name).copy_(getattr(cidx, name))
                if isinstance(cidx.valid, torch.Tensor):
                    held["cidx"].valid.copy_(cidx.valid)
            self.graph_captures_reused = getattr(self, "graph_captures_reused", 0) + 1
        else:
            # a capture that could only be matched by its ``extra`` OBJECT (no signature) is dead as soon as another batch arrives: free
            # it - graph, private memory pool, static buffers - BEFORE capturing, so that at most one such capture is ever resident
            if sig is None:
                dead = [h for h in pool if h["sig"] is None]
                if dead:
                    pool[:] = [h for h in pool if h["sig"] is not None]
                    if getattr(self, "_graph_cache", None) in dead:
                        self._graph_cache = None
                    self.last_graph = None
                    del dead
            # PRIVATE static buffers: the caller's tensors are never aliased by a capture, and the result is always a copy (below)
            xs, cs = x.clone(), clean.clone()
            u8 = ops.alloc_u8(x.shape[0], rows, x.shape[3], x.device) if any_export else None
            kw = {"clean_index": cidx} if cidx is not None else {}
            # labels shared by signature are captured through a private copy (later batches' tensors are copied INTO it: the caller's stay untouched)
            static_extra = adapter.graph_clone_extra(batch.extra) if sig is not None else batch.extra

            def iteration():
                loss, grad = adapter.loss_and_grad(xs, static_extra)
                ops.pgd_step(xs, grad.contiguous(), cs, sp, self.alpha, self.eps, out=xs, u8_out=u8, crop=(rows, cols) if u8 is not None else None, **kw)
                return loss

            keep = xs.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):                   # warm-up outside the capture (lazily built plans, LDS limits, MIOpen's solver search)
                for _ in range(2):
                    iteration()
            torch.cuda.current_stream(x.device).wait_stream(side)
            xs.copy_(keep)
            del keep
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss_buf = iteration().detach().reshape(()).clone()
            held = self._graph_cache = {"key": key, "adapter": adapter, "extra": static_extra, "sig": sig, "g": g, "x": xs, "clean": cs, "cidx": cidx, "u8": u8,
                                        "loss": loss_buf}
            pool.append(held)
            while len(pool) > self.max_graphs:
                pool.pop(0)                                  # the least recently used capture and its memory pool go
        g, u8, loss_buf = held["g"], held["u8"], held["loss"]
        for k in range(self.iters):
            g.replay()
            losses.append(loss_buf.clone())
            if self._wanted(k + 1):
                exporter.next_buffer().copy_(u8)
                exporter.submit(self._fan_out(k + 1, batch))
        self.last_graph = g
        # ... and nothing eager may start while the last replay is still in flight: without this second wait the NEXT batch's eager
        # preparation (upload, clean-image index) raced with the tail of the replays and the Stereo R-CNN-shaped step faulted - found with
        # tools/graph_replay_probe.py-style bisection: a wait after every batch makes it pass every time, none makes it fail every time
        torch.cuda.synchronize(xs.device)
        return xs.clone()                                   # always a copy: the static iterate is overwritten by the next batch's replays

    def run(self, loader, adapter, comm=None, debugnum=None):
        """Iterate a loader; with a Comm of world > 1 every rank takes the batches i % world == rank
        (no collective: attacked pairs are independent and write distinct files)."""
        comm = comm if comm is not None else Comm()
        done = 0
        for i, batch in _owned(loader, comm):
            if debugnum is not None and i * len(batch) > debugnum:       # pgd_attack.py:246-248 (quirk Q15)
                break
            self.run_batch(batch, adapter)
            done += len(batch)
        self.close()
        return done

    def close(self):
        if self.writer is not None:
            self.writer.close()
            self.writer = None


class PatchTrainer:
    """Universal round adversarial patch, pasted at a random position in both eyes and trained by
    DESCENDING the detection loss towards a fake target.

    Reference semantics (world 1, B = 1): per image, ``iters`` times { paste, fwd+bwd, patch -= clamp(500 *
    (gL_win + gR_win), +-eps) [, per-channel range clamp for Stereo R-CNN] }, the gradient buffer accumulating
    across the inner iterations because the scripts never zero ``img.grad`` (attack/DSGN/patch_attack.py:409-417;
    ``accumulate_grad=False`` turns that off).
    Data-parallel rule (B > 1 and/or world > 1): all images of a round are evaluated against the
    same patch snapshot, their clamped deltas are summed (in index order on a rank, all-reduce(SUM) across
    ranks) and applied at once; ``average=True`` divides the sum by the number of contributing pairs.
    """

    ALPHA = 1e3     # hard-coded in both scripts (patch_attack.py:279 / :101)

    def __init__(self, model_kind, ratio, eps, iters, epochs, out_root=".", seed=None, rng=None, comm=None,
                 accumulate_grad=True, average=False, ops=None, device=None):
        self.ops = ops if ops is not None else _default_ops()
        self.kind = model_kind
        if model_kind == "dsgn":
            self.space, self.prefix, self.shape = self.ops.Space.dsgn(), "dsgn", patchgeom.DSGN_SHAPE
            self.lo = self.hi = None                 # the DSGN patch is never range-clamped (quirk Q10)
        elif model_kind == "srcnn":
            self.space, self.prefix, self.shape = self.ops.Space.srcnn(), "stereo_rcnn", patchgeom.SRCNN_SHAPE
            self.lo, self.hi = self.space.lo, self.space.hi
        else:
            raise ValueError("model_kind must be 'dsgn' or 'srcnn'")
        self.ratio, self.eps, self.iters, self.epochs = ratio, float(eps), int(iters), int(epochs)
        self.out_root = out_root
        self.comm = comm if comm is not None else Comm()
        self.accumulate_grad, self.average = accumulate_grad, average
        self.device = device
        self.patch_dim, self.radius = patchgeom.init_patch_dims(self.shape[0], ratio)
        # one position stream per rank so that ranks do not paste at identical places
        s = None if seed is None else seed + 7919 * self.comm.rank
        self.sampler = patchgeom.CenterSampler(self.shape[0], self.shape[1], self.radius, "random", seed=s, rng=rng)
        self.patch = None
        self._xbuf = None
        self.positions = []                          # (epoch, round, name, center_l, center_r) log

    def init_patch(self, device):
        d0 = pixelio.patch_dir(self.prefix, self.ratio, 0, self.out_root)
        if self.comm.rank == 0:
            host, _ = pixelio.load_or_init_patch(d0, self.patch_dim, allow_resize=(self.kind == "dsgn"))
        self.comm.barrier()
        if self.comm.rank != 0:
            host, _ = pixelio.load_or_init_patch(d0, self.patch_dim, allow_resize=(self.kind == "dsgn"))
        self.patch = torch.from_numpy(host).to(device).contiguous()
        return self.patch

    # -- one round: B pairs on this rank against one patch snapshot ----------------------------------
    @under_solvers
    def train_batch(self, batch, adapter, contributes=True):
        ops, r = self.ops, self.radius
        dev = self.patch.device
        b = len(batch) if batch is not None else 0
        if batch is not None and tuple(batch.imgL.shape[2:]) != tuple(self.shape):
            batch, b = None, 0                       # wrong-shape pairs are skipped, patch_attack.py:318-320
        if not contributes:
            batch, b = None, 0
        x = None
        if b:
            x = _stack_on_device(batch, dev)
            cl, cr = zip(*[self.sampler.draw() for _ in range(b)])
            for i in range(b):
                self.positions.append((batch.names[i], list(cl[i]), list(cr[i])))
            if hasattr(adapter, "inject_fake_target"):
                adapter.inject_fake_target(batch.extra, cl, cr, r)       # patch_attack.py:336-354 / :187-207
            # one small upload per round: paste centres [2b,2] and update windows [b,3] are views of the same buffer
            flat = [v for c in cl for v in (c[0], c[1])] + [v for c in cr for v in (c[0], c[1])] + \
                   [v for l, rr in zip(cl, cr) for v in (l[0], l[1], rr[1])]
            cbuf = torch.tensor(flat, dtype=torch.int32).to(dev, non_blocking=True)
            centers2, centers3 = cbuf[:4 * b].view(2 * b, 2), cbuf[4 * b:].view(b, 3)
        single = (b == 1 and self.comm.world == 1 and not self.average)
        loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        gacc = None
        dd = 3 * self.patch_dim * self.patch_dim
        if not single and (self._xbuf is None or self._xbuf.device != dev):
            # the exchange buffer, allocated once: [3*D*D] delta + 1 element carrying the number of contributing pairs,
            # so that ``average`` needs no second collective
            self._xbuf = torch.zeros((dd + 1,), dtype=torch.float32, device=dev)
        for it in range(self.iters):
            if b:
                if single:                                                   # the reference's own sequence
                    ops.patch_paste(x[0:1], self.patch, cl[0][0], cl[0][1], r)    # :369-376
                    ops.patch_paste(x[1:2], self.patch, cr[0][0], cr[0][1], r)
                else:
                    ops.patch_paste_batch(x, self.patch, centers2, r)
                loss, grad = adapter.loss_and_grad(x, batch.extra)           # :384-412
                loss_sum = loss_sum + loss.detach().to(dev, torch.float32)   # :414, kept on the device
                gacc = grad if (gacc is None or not self.accumulate_grad) else gacc + grad
                gl, gr = split_eyes(gacc.contiguous())
            if single:
                ops.patch_update(self.patch, gl, gr, cl[0][0], cl[0][1], cr[0][1], r, self.eps, alpha=self.ALPHA,
                                 lo=self.lo, hi=self.hi)                     # :416-430 (+ :272-281)
            else:
                delta = self._xbuf[:dd].view(3, self.patch_dim, self.patch_dim)
                if b:
                    ops.patch_delta_batch(gl, gr, centers3, r, self.eps, alpha=self.ALPHA, out=delta)
                else:
                    delta.zero_()
                self._xbuf[dd:].fill_(float(b))
                self.comm.all_reduce_sum_(self._xbuf)                        # RCCL over xGMI (gloo in CPU tests): ONE message
                if self.average:
                    delta = delta / torch.clamp(self._xbuf[dd:], min=1.0)
                ops.patch_apply(self.patch, delta, lo=self.lo, hi=self.hi)
        return loss_sum, b

    def train(self, loader_factory, adapter, n_items=None, debugnum=None):
        """``loader_factory()`` -> an iterable of StereoBatch, re-created every epoch.  With world > 1 the
        batches are dealt round-robin; every rank runs the same number of rounds (a rank that has run out of
        batches contributes a zero delta) so the all-reduces stay matched."""
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.patch is None:
            self.init_patch(dev)
        comm = self.comm
        for epoch in range(self.epochs):
            if comm.rank == 0:
                print("Epoch {0}".format(epoch))                             # patch_attack.py:293
            loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
            loss_num = 0
            loader = loader_factory()
            if hasattr(loader, "__len__"):                                   # lazily: never hold more than the prefetch depth
                total = len(loader)
                if debugnum is not None:                                     # :314-316 (batch_idx * B > debugnum stops)
                    total = min(total, debugnum // max(1, getattr(loader, "batch", 1)) + 1)
                mine = (b for i, b in itertools.takewhile(lambda ib: ib[0] < total, _owned(loader, comm)))
            else:
                mine, total = [], 0
                for i, batch in enumerate(loader):
                    if debugnum is not None and i * len(batch) > debugnum:
                        break
                    total += 1
                    if i % comm.world == comm.rank:
                        mine.append(batch)
                mine = iter(mine)
            for rnd in range(comm.rounds(total)):
                batch = next(mine, None)
                ls, b = self.train_batch(batch, adapter, contributes=batch is not None)
                loss_sum = loss_sum + ls
                loss_num += 1 if b else 0
            stats = torch.stack([loss_sum, torch.tensor(float(loss_num), device=dev)])
            comm.all_reduce_sum_(stats)
            if comm.rank == 0 and float(stats[1]) > 0:
                print("Average loss for epoch{0}: {1}".format(str(epoch + 1), float(stats[0] / stats[1])))  # :434-435
        if comm.rank == 0:                                                   # :438-443
            pixelio.save_patch(pixelio.patch_dir(self.prefix, self.ratio, self.epochs, self.out_root),
                               self.patch.detach().cpu().numpy()[None] if self.patch.dim() == 3 else self.patch.detach().cpu().numpy())
        comm.barrier()
        return self.patch


class DetectUnderAttack:
    """Counterpart of the four ``predict_and_save_*`` scripts: run the detector without
    gradients on attacked stereo pairs and write one KITTI label file per image.

    mode 'pgd'   : the loader already yields attacked images (the ``*_pgd_iters_k`` folders swapped in for
                   ``image_2/3``, attack/DSGN/README.md:30,69) - nothing is modified here.
    mode 'patch' : the trained patch is pasted at a position drawn per image from the ``atk_mode`` column band
                   (attack/DSGN/predict_and_save_patch.py:361-391,430-459; Stereo R-CNN :82-112), wrong-shape
                   pairs are skipped (:423-425).
    ``detector.detect(x, extra)`` -> per pair a list of (cls, bbox[4], score, center[3], (h, w, l, ry)); the
    post-processing that produces those (FCOS3D post-processor + NMS, ``get_dimensions``) is upstream code.
    """

    def __init__(self, model_kind, mode, label_dir, patch=None, atk_mode="random", seed=None, rng=None, ops=None,
                 device=None):
        self.ops = ops if ops is not None else _default_ops()
        if mode not in ("pgd", "patch"):
            raise ValueError("mode must be 'pgd' or 'patch'")
        self.kind, self.mode, self.label_dir, self.device = model_kind, mode, label_dir, device
        self.shape = patchgeom.DSGN_SHAPE if model_kind == "dsgn" else patchgeom.SRCNN_SHAPE
        self.patch, self.sampler, self.positions = None, None, []
        if mode == "patch":
            if patch is None:
                raise Exception("Patch directory NOT found.")            # predict_and_save_patch.py:356
            self.patch = patch
            self.radius = int(patch.shape[-1]) // 2
            self.sampler = patchgeom.CenterSampler(self.shape[0], self.shape[1], self.radius, atk_mode, seed=seed, rng=rng)

    def prepare(self, batch):
        """the stacked batch [2B,3,H,W] the detector sees: as loaded (mode 'pgd'), or with the patch pasted at a fresh
        position per pair (mode 'patch'); None when the pair is skipped (wrong shape, predict_and_save_patch.py:423-425)"""
        dev = self.device if self.device is not None else batch.imgL.device
        x = _stack_on_device(batch, dev)
        if self.mode == "patch":
            if tuple(x.shape[2:]) != tuple(self.shape):
                return None
            b = len(batch)
            cl, cr = zip(*[self.sampler.draw() for _ in range(b)])
            self.positions.extend(zip(batch.names, cl, cr))
            centers = torch.tensor([[c[0], c[1]] for c in cl] + [[c[0], c[1]] for c in cr], dtype=torch.int32, device=dev)
            if self.patch.device != x.device:
                self.patch = self.patch.to(x.device)
            self.ops.patch_paste_batch(x, self.patch, centers, self.radius)
        return x

    @under_solvers
    def run(self, loader, detector, debugnum=None):
        written = 0
        for i, batch in enumerate(loader):
            if debugnum is not None and i * len(batch) > debugnum:
                break
            x = self.prepare(batch)
            if x is None:
                continue
            with torch.no_grad():
                dets = detector.detect(x, batch.extra)
            for name, d in zip(batch.names, dets):
                pixelio.write_kitti_labels(self.label_dir, int(os.path.splitext(name)[0]), d)
                written += 1
        return written
