"""ctypes binding of libpeaksegdisk_hip.so (the C ABI declared in include/peaksegdisk_hip.h).

The library is the only solver: importing this module fails loudly when it has not been
built (`python -c "import __graft_entry__ as g; g.build()"`), and every dynamic program
fails with ERROR_NO_HIP_DEVICE when no MI355X is visible.  There is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpeaksegdisk_hip.so")

ERROR_NO_HIP_DEVICE = 12
ERROR_DEVICE_SOLVER = 13
ERROR_DEVICE_MEMORY = 14


class PsdResult(ctypes.Structure):
    _fields_ = [
        ("status", ctypes.c_int),
        ("kernel_status", ctypes.c_int),
        ("n_segments", ctypes.c_int),
        ("n_peaks", ctypes.c_int),
        ("n_equality_constraints", ctypes.c_int),
        ("max_intervals", ctypes.c_int),
        ("total_intervals", ctypes.c_ulonglong),
        ("best_cost", ctypes.c_double),
        ("n_serial_env", ctypes.c_int),
        ("step_reached", ctypes.c_int),
        ("spill_steps", ctypes.c_int),
    ]


class PsdSearchRow(ctypes.Structure):
    _fields_ = [
        ("penalty_str", ctypes.c_char * 40),
        ("penalty", ctypes.c_double),
        ("total_loss", ctypes.c_double),
        ("peaks", ctypes.c_int),
        ("segments", ctypes.c_int),
        ("bases", ctypes.c_int),
        ("iteration", ctypes.c_int),
        ("under_peaks", ctypes.c_int),
        ("over_peaks", ctypes.c_int),
        ("cached", ctypes.c_int),
    ]


ERROR_SEARCH_ARGUMENTS = 15
ERROR_SEARCH_TOO_MANY_PEAKS = 16
SEARCH_NA = -2 ** 31


def declare(lib):
    """Attach argtypes/restypes for every symbol of include/peaksegdisk_hip.h."""
    c = ctypes
    lib.PeakSegFPOP_disk.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p]
    lib.PeakSegFPOP_disk.restype = c.c_int
    lib.PeakSegFPOP_disk_batch.argtypes = [
        c.c_int, c.POINTER(c.c_char_p), c.POINTER(c.c_char_p), c.POINTER(c.c_char_p),
        c.POINTER(c.c_int)]
    lib.PeakSegFPOP_disk_batch.restype = c.c_int
    lib.PeakSegFPOP_status_message.argtypes = [
        c.c_int, c.c_char_p, c.c_char_p, c.c_char_p, c.c_char_p, c.c_size_t]
    lib.PeakSegFPOP_status_message.restype = c.c_char_p
    lib.peakseg_hip_set_print.argtypes = [c.c_void_p]
    lib.peakseg_hip_set_print.restype = None
    lib.peakseg_hip_device_count.argtypes = []
    lib.peakseg_hip_device_count.restype = c.c_int
    lib.peakseg_hip_last_error.argtypes = []
    lib.peakseg_hip_last_error.restype = c.c_char_p
    lib.peakseg_hip_problem_set_create.argtypes = [
        c.c_int, c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_void_p), c.POINTER(c.c_void_p),
        c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_double), c.c_ulonglong,
        c.POINTER(c.c_void_p)]
    lib.peakseg_hip_problem_set_create.restype = c.c_int
    lib.peakseg_hip_problem_set_solve.argtypes = [
        c.c_void_p, c.POINTER(c.c_float), c.POINTER(c.c_float)]
    lib.peakseg_hip_problem_set_solve.restype = c.c_int
    lib.peakseg_hip_problem_set_result.argtypes = [c.c_void_p, c.c_int, c.POINTER(PsdResult)]
    lib.peakseg_hip_problem_set_result.restype = c.c_int
    lib.peakseg_hip_problem_set_segments.argtypes = [
        c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p]
    lib.peakseg_hip_problem_set_segments.restype = c.c_int
    lib.peakseg_hip_problem_set_export_db.argtypes = [
        c.c_void_p, c.c_int, c.c_void_p, c.c_char_p]
    lib.peakseg_hip_problem_set_export_db.restype = c.c_int
    lib.peakseg_hip_problem_set_bytes.argtypes = [c.c_void_p]
    lib.peakseg_hip_problem_set_bytes.restype = c.c_ulonglong
    lib.peakseg_hip_problem_set_kernel_build.argtypes = [c.c_void_p]
    lib.peakseg_hip_problem_set_kernel_build.restype = c.c_char_p
    lib.peakseg_hip_problem_set_destroy.argtypes = [c.c_void_p]
    lib.peakseg_hip_problem_set_destroy.restype = None
    lib.peakseg_hip_parse_probe.argtypes = [c.c_char_p, c.c_int, c.POINTER(c.c_int),
                                            c.POINTER(c.c_ulonglong)]
    lib.peakseg_hip_parse_probe.restype = c.c_int
    lib.peakseg_hip_problem_set_profile.argtypes = [c.c_void_p, c.c_int, c.c_void_p]
    lib.peakseg_hip_problem_set_profile.restype = c.c_int
    lib.peakseg_hip_math_probe.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_void_p]
    lib.peakseg_hip_math_probe.restype = c.c_int
    lib.PeakSegFPOP_dir_batch.argtypes = [
        c.c_int, c.POINTER(c.c_char_p), c.POINTER(c.c_char_p), c.POINTER(c.c_int),
        c.POINTER(c.c_int)]
    lib.PeakSegFPOP_dir_batch.restype = c.c_int
    lib.PeakSegFPOP_sequential_search.argtypes = [
        c.c_char_p, c.c_int, c.c_int, c.c_int, c.POINTER(PsdSearchRow), c.POINTER(c.c_int),
        c.POINTER(c.c_int)]
    lib.PeakSegFPOP_sequential_search.restype = c.c_int
    lib.PeakSegFPOP_sequential_search_batch.argtypes = [
        c.c_int, c.POINTER(c.c_char_p), c.POINTER(c.c_int), c.c_int, c.c_int,
        c.POINTER(PsdSearchRow), c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int)]
    lib.PeakSegFPOP_sequential_search_batch.restype = c.c_int
    lib.peakseg_hip_problem_set_set_penalty.argtypes = [c.c_void_p, c.c_int, c.c_double]
    lib.peakseg_hip_problem_set_set_penalty.restype = c.c_int
    lib.peakseg_hip_problem_set_arena_bytes_used.argtypes = [c.c_void_p]
    lib.peakseg_hip_problem_set_arena_bytes_used.restype = c.c_ulonglong
    lib.peakseg_hip_problem_set_checkpoint_interval.argtypes = [c.c_void_p]
    lib.peakseg_hip_problem_set_checkpoint_interval.restype = c.c_int
    lib.peakseg_hip_paste_double.argtypes = [c.c_double, c.c_char_p, c.c_size_t]
    lib.peakseg_hip_paste_double.restype = c.c_int
    lib.peakseg_hip_problem_set_solve_stats.argtypes = [
        c.c_void_p, c.POINTER(c.c_int), c.POINTER(c.c_ulonglong)]
    lib.peakseg_hip_problem_set_solve_stats.restype = c.c_int
    lib.peakseg_hip_device_clock_khz.argtypes = [c.c_int]
    lib.peakseg_hip_device_clock_khz.restype = c.c_int
    lib.peakseg_hip_problem_set_park_stats.argtypes = [
        c.c_void_p, c.POINTER(c.c_int), c.POINTER(c.c_ulonglong)]
    lib.peakseg_hip_problem_set_park_stats.restype = c.c_int
    lib.peakseg_hip_problem_set_pack_tables.argtypes = [
        c.c_void_p, c.c_void_p, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p)]
    lib.peakseg_hip_problem_set_pack_tables.restype = c.c_longlong
    lib.peakseg_hip_problem_set_packed_download.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p]
    lib.peakseg_hip_problem_set_packed_download.restype = c.c_int
    lib.peakseg_hip_problem_set_cycles.argtypes = [c.c_void_p, c.c_int]
    lib.peakseg_hip_problem_set_cycles.restype = c.c_longlong
    lib.peakseg_hip_measured_rates.argtypes = [c.POINTER(c.c_double), c.POINTER(c.c_double)]
    lib.peakseg_hip_measured_rates.restype = None
    lib.peakseg_hip_spin_limit.argtypes = []
    lib.peakseg_hip_spin_limit.restype = c.c_longlong
    lib.peakseg_hip_problem_set_max_spin.argtypes = [c.c_void_p, c.c_int]
    lib.peakseg_hip_problem_set_max_spin.restype = c.c_int
    lib.peakseg_hip_last_warning.argtypes = []
    lib.peakseg_hip_last_warning.restype = c.c_char_p
    lib.peakseg_hip_problem_set_arena_stats.argtypes = [
        c.c_void_p, c.POINTER(c.c_ulonglong), c.POINTER(c.c_int), c.POINTER(c.c_int)]
    lib.peakseg_hip_problem_set_arena_stats.restype = c.c_int
    return lib


EXPORTED_SYMBOLS = [
    "PeakSegFPOP_disk", "PeakSegFPOP_disk_batch", "PeakSegFPOP_status_message",
    "peakseg_hip_set_print", "peakseg_hip_device_count", "peakseg_hip_last_error",
    "peakseg_hip_problem_set_create", "peakseg_hip_problem_set_solve",
    "peakseg_hip_problem_set_result", "peakseg_hip_problem_set_segments",
    "peakseg_hip_problem_set_export_db", "peakseg_hip_problem_set_bytes",
    "peakseg_hip_problem_set_destroy", "peakseg_hip_math_probe", "peakseg_hip_parse_probe",
    "peakseg_hip_problem_set_profile", "peakseg_hip_problem_set_kernel_build",
    "PeakSegFPOP_dir_batch", "PeakSegFPOP_sequential_search",
    "PeakSegFPOP_sequential_search_batch",
    "peakseg_hip_problem_set_set_penalty", "peakseg_hip_problem_set_arena_bytes_used",
    "peakseg_hip_paste_double", "peakseg_hip_problem_set_checkpoint_interval",
    "peakseg_hip_problem_set_solve_stats", "peakseg_hip_device_clock_khz",
    "peakseg_hip_last_warning", "peakseg_hip_problem_set_arena_stats",
    "peakseg_hip_problem_set_park_stats", "peakseg_hip_problem_set_pack_tables",
    "peakseg_hip_problem_set_packed_download", "peakseg_hip_problem_set_cycles",
    "peakseg_hip_measured_rates", "peakseg_hip_spin_limit", "peakseg_hip_problem_set_max_spin",
]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "%s is missing: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()'). "
        "peaksegdisk_amd has no CPU fallback." % LIB_PATH)

lib = declare(ctypes.CDLL(LIB_PATH))


def last_error():
    return lib.peakseg_hip_last_error().decode(errors="replace")


def status_message(status, bedgraph, penalty, db):
    buf = ctypes.create_string_buffer(4096)
    lib.PeakSegFPOP_status_message(
        status, os.fsencode(bedgraph), penalty.encode(), os.fsencode(db), buf, len(buf))
    return buf.value.decode(errors="replace")
