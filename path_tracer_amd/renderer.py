"""HipRenderer — Python mirror of the reference's ``VulkanRenderer`` interface
(reference include/VulkanRenderer.h:73-97) over the C-ABI of include/pt_hip.h.

Method names and argument meaning follow the reference class:
``initialize`` / ``updateScene`` / ``render`` / ``present`` /
``getShaderPathToIndexMap`` / ``shutdown``; ``readback*`` replace the
swapchain (the frame is returned as a numpy array instead of shown).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from . import types as T


def _ptr(a: Optional[np.ndarray]):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data


class HipRenderer:
    def __init__(self):
        self._h = C.c_void_p()
        self._lib = None
        self.width = 0
        self.height = 0
        self._honour_max_depth = False
        self._kept = {}

    # -- VulkanRenderer::initialize(window, width, height); no window here
    def initialize(self, width: int, height: int, device: int = 0) -> bool:
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.pt_create(width, height, device, C.byref(h))
        if rc != 0:
            # the reference reports failure as `false` + a message on stderr
            import sys

            print("HipRenderer.initialize failed:", self._lib.pt_last_error().decode(), file=sys.stderr)
            return False
        self._h = h
        self.width, self.height = width, height
        self.setOption(7, 1 if self._honour_max_depth else 0)
        return True

    # The reference shader ignores PushConstants.maxDepth and always runs 10 bounces
    # (reference shaders/raytrace.comp:603-605), and so does render() by default.
    # honour_max_depth = True is an extension: maxDepth > 0 becomes the bounce limit.
    @property
    def honour_max_depth(self) -> bool:
        return self._honour_max_depth

    @honour_max_depth.setter
    def honour_max_depth(self, on: bool) -> None:
        self._honour_max_depth = bool(on)
        if self._h:
            self.setOption(7, 1 if on else 0)  # PT_OPT_HONOUR_MAX_DEPTH

    # -- VulkanRenderer::shutdown (idempotent)
    def shutdown(self) -> None:
        if self._h:
            self._lib.pt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    # -- VulkanRenderer::getShaderPathToIndexMap
    def getShaderPathToIndexMap(self) -> Dict[str, int]:
        lib = self._lib or _lib.load()
        return {lib.pt_shader_path(i).decode(): i + 1 for i in range(lib.pt_shader_count())}

    # -- VulkanRenderer::updateScene(spheres, ellipsoids, materials, lights, volumes, voxelData = {})
    def updateScene(self, spheres, ellipsoids, materials, lights, volumes, voxelData=None) -> None:
        # An empty (or None) array keeps what the previous call supplied for it: the reference skips
        # the copy of an empty vector and its buffer keeps the old records
        # (reference src/VulkanRenderer.cpp:801,850,876,910,936).
        arrs = []
        for key, a, dt in (("s", spheres, T.GPU_SPHERE), ("e", ellipsoids, T.GPU_ELLIPSOID), ("m", materials, T.GPU_MATERIAL),
                           ("l", lights, T.GPU_LIGHT), ("v", volumes, T.GPU_VOLUME)):
            a = np.zeros(0, dt) if a is None else np.ascontiguousarray(a, dtype=dt)
            if len(a):
                self._kept[key] = a
            arrs.append(self._kept.get(key, a))
        vox = None if voxelData is None else np.ascontiguousarray(voxelData, dtype=np.uint8)
        _lib.check(self._lib.pt_update_scene(
            self._h, _ptr(arrs[0]), len(arrs[0]), _ptr(arrs[1]), len(arrs[1]), _ptr(arrs[2]), len(arrs[2]),
            _ptr(arrs[3]), len(arrs[3]), _ptr(arrs[4]), len(arrs[4]), _ptr(vox), 0 if vox is None else vox.size))

    def _pc(self, pushConstants) -> np.ndarray:
        return np.array(pushConstants, dtype=T.PUSH_CONSTANTS, copy=True).reshape(())

    # -- VulkanRenderer::render(pushConstants): asynchronous
    def render(self, pushConstants) -> None:
        pc = self._pc(pushConstants)
        _lib.check(self._lib.pt_render(self._h, pc.ctypes.data))

    # -- VulkanRenderer::present(): wait for the frame
    def present(self) -> None:
        _lib.check(self._lib.pt_present(self._h))

    # ---- beyond the reference interface ----
    def setScene(self, scene, honour_max_depth: bool = False) -> None:
        """Upload a ``scenes.Scene``.  Like ``updateScene`` and the C default this keeps the
        reference's fixed 10 bounces (reference shaders/raytrace.comp:603) unless
        ``honour_max_depth=True``, which makes the scene's stated depth (``scene.max_depth``,
        sent as ``maxDepth``) the bounce limit: the named configurations are drawn that way."""
        self.honour_max_depth = honour_max_depth
        self.updateScene(scene.spheres, scene.ellipsoids, scene.materials, scene.lights, scene.volumes, scene.voxels)

    def resetVoxels(self) -> None:
        _lib.check(self._lib.pt_reset_voxels(self._h))

    def localRows(self) -> int:
        return self._lib.pt_local_rows(self._h)

    def readbackRGBA8(self) -> np.ndarray:
        out = np.empty((self.localRows(), self.width, 4), dtype=np.uint8)
        _lib.check(self._lib.pt_readback_rgba8(self._h, out.ctypes.data, out.nbytes))
        return out

    def readbackF32(self) -> np.ndarray:
        out = np.empty((self.localRows(), self.width, 4), dtype=np.float32)
        _lib.check(self._lib.pt_readback_f32(self._h, out.ctypes.data, out.nbytes))
        return out

    def renderCounted(self, pushConstants) -> np.ndarray:
        pc = self._pc(pushConstants)
        cnt = np.zeros(len(T.COUNTER_NAMES), dtype=np.uint64)
        _lib.check(self._lib.pt_render_counted(self._h, pc.ctypes.data, cnt.ctypes.data))
        return cnt

    def streamConcurrency(self) -> int:
        """Slots' streams the device runs side by side (measured; see pt_stream_concurrency)."""
        n = C.c_int()
        _lib.check(self._lib.pt_stream_concurrency(self._h, C.byref(n)))
        return n.value

    def waveTimeline(self, pushConstants) -> np.ndarray:
        """[waves, 4] uint64: start / tickets-drained / exit clock stamps of every wave of one frame, and its pass
        count (low 32 bits of the fourth word; high 32 bits: passes made when it found the tickets drained)."""
        pc = self._pc(pushConstants)
        out = np.zeros((16384, 4), dtype=np.uint64)
        n = C.c_int()
        _lib.check(self._lib.pt_wave_timeline(self._h, pc.ctypes.data, out.ctypes.data, len(out), C.byref(n)))
        return out[:n.value]

    def renderSamples(self, pushConstants, first_sample: int = 0, num_samples: int = 1, keep: bool = False) -> None:
        """Mean of sub-pixel samples first_sample .. first_sample + num_samples - 1 (sample 0 is the
        reference's pixel-corner sample); keep=True continues the previous call's sum."""
        pc = self._pc(pushConstants)
        _lib.check(self._lib.pt_render_samples(self._h, pc.ctypes.data, first_sample, num_samples, 1 if keep else 0))

    def renderStatic(self, pushConstants, rgba8_ptr: int, f32_ptr: Optional[int] = None,
                     stream_ptr: Optional[int] = None) -> None:
        """pt_render_static: the frame into the caller's device buffers on the caller's stream, allocating
        nothing (the frame's scratch is a static device array shared by every renderer on the GPU)."""
        pc = self._pc(pushConstants)
        _lib.check(self._lib.pt_render_static(self._h, pc.ctypes.data, rgba8_ptr, f32_ptr, stream_ptr))

    def setRowBands(self, band_rows: int, band_index: int, band_count: int) -> None:
        _lib.check(self._lib.pt_set_row_bands(self._h, band_rows, band_index, band_count))

    def assembleBands(self, src_ptr: int, dst_ptr: int, elem_bytes: int, band_rows: int, band_count: int,
                      stride_rows: int) -> None:
        _lib.check(self._lib.pt_assemble_bands(self._h, src_ptr, dst_ptr, elem_bytes, band_rows, band_count,
                                               stride_rows))

    # ---- the gather over RCCL behind the C-ABI (no torch involved) ----
    @staticmethod
    def commUniqueId() -> bytes:
        """Rank 0: the bytes that identify a new communicator; hand them to every other rank."""
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        _lib.check(lib.pt_comm_unique_id(buf, 128))
        return buf.raw

    def commCreate(self, unique_id: bytes, rank: int, world: int, device: int = 0):
        """Collective over all ranks: joins the communicator and switches this renderer to rank's row bands."""
        comm = C.c_void_p()
        _lib.check(self._lib.pt_comm_create(unique_id, len(unique_id), rank, world, device, C.byref(comm)))
        return comm

    def commDestroy(self, comm) -> None:
        if comm:
            self._lib.pt_comm_destroy(comm)

    def gatherBands(self, comm, elem_bytes: int = 4, dst_ptr: Optional[int] = None) -> None:
        _lib.check(self._lib.pt_gather_bands(self._h, comm, elem_bytes, dst_ptr))

    def readbackAssembled(self, elem_bytes: int = 4) -> np.ndarray:
        """Rank 0: the frame pt_gather_bands assembled into the renderer's own buffer."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8 if elem_bytes == 4 else np.float32)
        _lib.check(self._lib.pt_readback_assembled(self._h, out.ctypes.data, out.nbytes))
        return out

    def setOutput(self, rgba8_ptr: Optional[int], f32_ptr: Optional[int] = None) -> None:
        _lib.check(self._lib.pt_set_output(self._h, rgba8_ptr, f32_ptr))

    def setStream(self, stream_ptr: Optional[int]) -> None:
        _lib.check(self._lib.pt_set_stream(self._h, stream_ptr))

    def setSlot(self, slot: int, stream_ptr: Optional[int], rgba8_ptr: Optional[int], f32_ptr: Optional[int] = None) -> None:
        _lib.check(self._lib.pt_set_slot(self._h, slot, stream_ptr, rgba8_ptr, f32_ptr))

    def lastSlot(self) -> int:
        return self._lib.pt_last_slot(self._h)

    def setOption(self, option: int, value: int) -> None:
        _lib.check(self._lib.pt_set_option(self._h, option, value))

    def timerBegin(self) -> None:
        _lib.check(self._lib.pt_timer_begin(self._h))

    def timerEnd(self) -> None:
        _lib.check(self._lib.pt_timer_end(self._h))

    def timerElapsedMs(self) -> float:
        ms = C.c_float()
        _lib.check(self._lib.pt_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def deviceInfo(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        _lib.check(self._lib.pt_device_info(self._h, name, 256, C.byref(cu)))
        return name.value.decode(), cu.value

    def lastLaunch(self):
        b, t, l = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.pt_last_launch(self._h, C.byref(b), C.byref(t), C.byref(l)))
        return b.value, t.value, l.value

    def devicePointers(self):
        return self._lib.pt_device_rgba8(self._h), self._lib.pt_device_f32(self._h)


def render_scene(scene, device: int = 0, counted: bool = False, reference_bounce_limit: bool = False):
    """Convenience for the named configurations, which state a bounce depth: one frame of ``scene``
    with ``scene.max_depth`` honoured -> (rgba8, f32[, counters]).  ``reference_bounce_limit=True``
    draws it the way HipRenderer does by default (10 bounces, maxDepth ignored)."""
    r = HipRenderer()
    if not r.initialize(scene.width, scene.height, device):
        raise RuntimeError("HipRenderer.initialize failed")
    try:
        r.setScene(scene, honour_max_depth=not reference_bounce_limit)
        pc = scene.push_constants()
        r.render(pc)
        r.present()
        out = (r.readbackRGBA8(), r.readbackF32())
        if counted:
            out = out + (r.renderCounted(pc),)
        return out
    finally:
        r.shutdown()
