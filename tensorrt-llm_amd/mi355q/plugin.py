"""Plugin objects over the C ABI: what a TensorRT execution context does with a plugin
(create -> configure -> getWorkspaceSize -> enqueue), driven eagerly on torch-ROCm tensors.

torch is plumbing here (device memory, streams); all arithmetic happens inside libmi355q.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

_TORCH2TRT = {
    torch.float32: L.FLOAT, torch.float16: L.HALF, torch.int8: L.INT8, torch.int32: L.INT32,
    torch.bool: L.BOOL, torch.uint8: L.UINT8, torch.bfloat16: L.BF16, torch.int64: L.INT64,
}
if hasattr(torch, "float8_e4m3fn"):
    _TORCH2TRT[torch.float8_e4m3fn] = L.FP8
_TRT2TORCH = {v: k for k, v in _TORCH2TRT.items()}


def torch_dtype_to_trt(dt):
    return _TORCH2TRT[dt]


def trt_dtype_to_torch(code):
    return _TRT2TORCH[code]


def str_dtype_to_trt(s):
    return {"float16": L.HALF, "bfloat16": L.BF16, "float32": L.FLOAT, "int32": L.INT32, "int8": L.INT8,
            "fp8": L.FP8}[s]


def current_stream_ptr():
    """hipStream_t of torch's current stream (NULL stream when no GPU is present: host-only checks)."""
    if not torch.cuda.is_available():
        return C.c_void_p(0)
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _field_array(fields):
    """fields: list of (name, value, FIELD_* type).  Returns (ctypes array, keepalive)."""
    arr = (L.PluginField * max(1, len(fields)))()
    keep = []
    np_t = {L.FIELD_INT8: np.int8, L.FIELD_INT16: np.int16, L.FIELD_INT32: np.int32,
            L.FIELD_FLOAT32: np.float32, L.FIELD_FLOAT64: np.float64}
    for i, (name, value, ftype) in enumerate(fields):
        buf = np.array(list(value) if isinstance(value, (list, tuple)) else [value], dtype=np_t[ftype])
        nm = name.encode()
        keep += [buf, nm]
        arr[i].name = nm
        arr[i].data = buf.ctypes.data
        arr[i].type = ftype
        arr[i].length = buf.size
    return arr, keep


class Plugin:
    """One plugin instance (IPluginV2DynamicExt).  Inputs are torch tensors; tensors the reference
    keeps on the host (GPTAttention HOST_* entries) must be CPU tensors."""

    def __init__(self, name, fields=(), _handle=None):
        self.name = name
        if _handle is None:
            arr, keep = _field_array(list(fields))
            h = C.c_void_p()
            L.check(L.lib().mi355q_plugin_create(name.encode(), arr, len(fields), C.byref(h)))
            _handle = h
        self._h = _handle
        self._workspace = None

    def __del__(self):
        try:
            if self._h:
                L.lib().mi355q_plugin_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- creator / serialisation
    @classmethod
    def deserialize(cls, name, blob):
        h = C.c_void_p()
        buf = (C.c_char * max(1, len(blob))).from_buffer_copy(blob if len(blob) else b"\0")
        L.check(L.lib().mi355q_plugin_deserialize(name.encode(), buf, len(blob), C.byref(h)))
        return cls(name, _handle=h)

    def serialize(self):
        n = L.lib().mi355q_plugin_get_serialization_size(self._h)
        buf = (C.c_char * max(1, n))()
        L.check(L.lib().mi355q_plugin_serialize(self._h, buf))
        return bytes(buf[:n])

    def clone(self):
        h = C.c_void_p()
        L.check(L.lib().mi355q_plugin_clone(self._h, C.byref(h)))
        return Plugin(self.name, _handle=h)

    @property
    def plugin_type(self):
        return L.lib().mi355q_plugin_type(self._h).decode()

    @property
    def nb_outputs(self):
        return L.lib().mi355q_plugin_nb_outputs(self._h)

    # -- shape / type inference
    @staticmethod
    def _descs(tensors):
        arr = (L.TensorDesc * max(1, len(tensors)))()
        for i, t in enumerate(tensors):
            arr[i] = L.TensorDesc.of(list(t.shape), torch_dtype_to_trt(t.dtype))
        return arr

    def get_output_dimensions(self, index, input_shapes):
        dims = (L.Dims * len(input_shapes))(*[L.Dims.of(s) for s in input_shapes])
        out = L.Dims()
        L.check(L.lib().mi355q_plugin_get_output_dimensions(self._h, index, dims, len(input_shapes), C.byref(out)))
        return out.tolist()

    def get_output_data_type(self, index, input_types):
        arr = (C.c_int32 * len(input_types))(*input_types)
        r = L.lib().mi355q_plugin_get_output_data_type(self._h, index, arr, len(input_types))
        if r < 0:
            L.check(r)
        return r

    def supports_format_combination(self, pos, descs, nb_inputs, nb_outputs):
        r = L.lib().mi355q_plugin_supports_format_combination(self._h, pos, descs, nb_inputs, nb_outputs)
        if r < 0:
            L.check(r)
        return bool(r)

    def configure(self, inputs, min_shapes=None, max_shapes=None):
        n = len(inputs)
        arr = (L.DynamicTensorDesc * n)()
        for i, t in enumerate(inputs):
            arr[i].desc = L.TensorDesc.of(list(t.shape), torch_dtype_to_trt(t.dtype))
            arr[i].min = L.Dims.of(list(min_shapes[i]) if min_shapes else list(t.shape))
            arr[i].max = L.Dims.of(list(max_shapes[i]) if max_shapes else list(t.shape))
        L.check(L.lib().mi355q_plugin_configure(self._h, arr, n, None, self.nb_outputs))

    def get_workspace_size(self, inputs, outputs):
        return L.lib().mi355q_plugin_get_workspace_size(
            self._h, self._descs(inputs), len(inputs), self._descs(outputs), len(outputs))

    # -- execution
    def enqueue(self, inputs, outputs, workspace=None):
        in_desc, out_desc = self._descs(inputs), self._descs(outputs)
        in_ptrs = (C.c_void_p * max(1, len(inputs)))(*[t.data_ptr() for t in inputs])
        out_ptrs = (C.c_void_p * max(1, len(outputs)))(*[t.data_ptr() for t in outputs])
        ws = C.c_void_p(workspace.data_ptr()) if workspace is not None and workspace.numel() else None
        rc = L.lib().mi355q_plugin_enqueue(self._h, in_desc, out_desc, in_ptrs, out_ptrs, ws, current_stream_ptr())
        L.check(rc)

    def __call__(self, *inputs, device=None):
        """configure + allocate outputs/workspace (torch) + enqueue; returns the output tensor(s)."""
        inputs = list(inputs)
        dev = device or next(t.device for t in inputs if t.is_cuda)
        shapes = [list(t.shape) for t in inputs]
        types = [torch_dtype_to_trt(t.dtype) for t in inputs]
        outs = []
        for i in range(self.nb_outputs):
            shp = self.get_output_dimensions(i, shapes)
            dt = trt_dtype_to_torch(self.get_output_data_type(i, types))
            outs.append(torch.empty(shp, dtype=dt, device=dev))
        self.configure(inputs)
        ws_bytes = self.get_workspace_size(inputs, outs)
        if ws_bytes and (self._workspace is None or self._workspace.numel() < ws_bytes
                         or self._workspace.device != dev):
            self._workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        self.enqueue(inputs, outs, self._workspace if ws_bytes else None)
        return outs[0] if len(outs) == 1 else tuple(outs)


def get_plugin_names():
    n = C.c_int32()
    arr = L.lib().mi355q_get_plugin_names(C.byref(n))
    return [arr[i].decode() for i in range(n.value)]
