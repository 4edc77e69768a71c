"""ctypes binding of include/vkimg.h (libvkimg_hip.so).  No CPU fallback: if the
HIP library is missing or a call fails, the caller gets an exception."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VKIMG_LIB") or os.path.join(HERE, "libvkimg_hip.so")  # (VKIMG_LIB: A/B timing of experimental builds)

VK_OK, VK_EINVAL, VK_EHIP, VK_ENOMAP, VK_EFORMAT, VK_ENOMEM = 0, 1, 2, 3, 4, 5
VK_ST_BAD_START, VK_ST_BAD_PHASE = 1, 2
VK_GZ_BAD_HEADER, VK_GZ_BAD_DATA, VK_GZ_TRUNCATED, VK_GZ_OVERFLOW, VK_GZ_BAD_SIZE, VK_GZ_BAD_CRC = 1, 2, 4, 8, 16, 32

# every symbol include/vkimg.h declares
SYMBOLS = ("vk_abi_version", "vk_strerror", "vk_last_hip_error", "vk_ctx_create", "vk_ctx_destroy",
           "vk_ctx_sync", "vk_set_mapping", "vk_count_device", "vk_image_device",
           "vk_fastq_to_image_device", "vk_count_host", "vk_image_host", "vk_synth_fastq_device", "vk_remap_host", "vk_preprocess_device",
           "vk_last_count_launch", "vk_count_sampled_device", "vk_inflate_device", "vk_upload_mapped", "vk_host_register", "vk_host_unregister",
           "vk_synth_shaped_lengths", "vk_synth_shaped_device", "vk_last_count_general", "vk_read_index_device", "vk_count_index_device")

_lib = None


class VkError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VkError(VK_EHIP, f"HIP extension not built: {LIB_PATH} is missing "
                               "(run `python -m varkoder_amd.build`); there is no CPU fallback")
    # PyTorch carries its own HIP runtime (libamdhip64 under torch/lib): it must be the process's one before this library is
    # loaded, else the loader resolves the library's dependency to the system copy, PyTorch later brings its own, and a
    # context created here on a stream PyTorch made fails with a HIP runtime error (seen: `build()` then `smoke()` in one process).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32p, u64p, u8p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    L.vk_abi_version.restype = C.c_int
    L.vk_strerror.restype = C.c_char_p
    L.vk_strerror.argtypes = [C.c_int]
    L.vk_last_hip_error.restype = C.c_char_p
    L.vk_last_hip_error.argtypes = [vp]
    L.vk_ctx_create.argtypes = [C.c_int, vp, C.c_int, C.POINTER(vp)]
    L.vk_ctx_destroy.restype = None
    L.vk_ctx_destroy.argtypes = [vp]
    L.vk_ctx_sync.argtypes = [vp]
    L.vk_set_mapping.argtypes = [vp, C.c_int, u32p, C.c_uint32]
    L.vk_count_device.argtypes = [vp, vp, u64p, u64p, C.c_uint32, C.c_int, C.c_uint32, vp, vp]
    L.vk_count_sampled_device.argtypes = [vp, vp, u64p, u64p, C.c_uint32, C.c_int, C.c_uint32, u64p, u64p, vp, vp, vp]
    L.vk_image_device.argtypes = [vp, vp, C.c_uint32, C.c_int, vp]
    L.vk_fastq_to_image_device.argtypes = [vp, vp, u64p, u64p, C.c_uint32, C.c_int, C.c_uint32, vp, vp, vp]
    L.vk_count_host.argtypes = [vp, vp, C.c_size_t, C.c_int, u32p, u32p]
    L.vk_image_host.argtypes = [vp, u32p, C.c_int, u8p]
    L.vk_synth_fastq_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int]
    L.vk_synth_shaped_lengths.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, u64p]
    L.vk_synth_shaped_device.argtypes = [vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
    L.vk_read_index_device.argtypes = [vp, vp, u64p, u64p, C.c_uint32, C.c_uint32, u64p, u32p]
    L.vk_count_index_device.argtypes = [vp, vp, u64p, u64p, C.c_uint32, C.c_int, C.c_uint32, vp, vp, u64p, u32p]
    L.vk_last_count_general.argtypes = [vp, u64p, u64p]
    L.vk_last_count_launch.argtypes = [vp, u32p, u32p, u32p]
    L.vk_preprocess_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32, C.c_float, C.c_float, vp]
    L.vk_remap_host.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_int, vp]
    L.vk_inflate_device.argtypes = [vp, vp, u64p, u64p, C.c_uint32, vp, u64p, u64p, u64p, u32p]
    L.vk_upload_mapped.argtypes = [vp, vp, u64p, C.POINTER(vp), u64p, u8p, C.c_uint32, u32p]
    L.vk_host_register.argtypes = [vp, vp, C.c_uint64]
    L.vk_host_unregister.argtypes = [vp, vp]
    _lib = L
    return L


def check(ctx, status, what):
    if status == VK_OK:
        return
    L = lib()
    msg = L.vk_strerror(status).decode()
    if status == VK_EHIP and ctx:
        msg += ": " + L.vk_last_hip_error(ctx).decode()
    raise VkError(status, f"{what}: {msg}")
