"""ctypes binding of libmcgp_hip.so (C ABI: include/mcgp.h).

There is no CPU fallback: if the library is missing it is built with hipcc, and
if that fails, or no HIP device is visible, the error is raised to the caller.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, 'csrc')
LIB_PATH = os.path.join(_PKG, 'libmcgp_hip.so')

MAX_CARS = 32
MAX_LAPS = 1000
ABI_VERSION = 2

COMPOUNDS = ('SOFT', 'MEDIUM', 'HARD', 'INTERMEDIATE', 'WET')
COMPOUND_ID = {c: i for i, c in enumerate(COMPOUNDS)}
TRACK_ID = {'dry': 0, 'damp': 1, 'wet': 2}


class McgpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'mcgp error {code}: {msg}')
        self.code = code


class McgpConfig(C.Structure):
    _fields_ = [
        ('total_laps', C.c_int32), ('track_condition', C.c_int32),
        ('pit_loss', C.c_double), ('overtake_delta', C.c_double),
        ('sc_probability', C.c_double), ('vsc_probability', C.c_double),
        ('red_flag_probability', C.c_double), ('drs_delta', C.c_double),
        ('dirty_air_threshold', C.c_double), ('dirty_air_penalty', C.c_double),
        ('comp_pace_delta', C.c_double * 5), ('comp_deg_rate', C.c_double * 5),
        ('comp_optimal_laps', C.c_int32 * 5),
        ('pop_soft_hard', C.c_int32), ('pop_medium_hard', C.c_int32),
        ('deviates', C.c_int32),          # 0: 32-bit deviates (default), 1: the reference's 53-bit / binary64 deviates
    ]


class McgpDrivers(C.Structure):
    _fields_ = [(k, C.POINTER(C.c_double)) for k in
                ('base_pace', 'tire_deg', 'tire_deg_pit', 'variance', 'team_dnf', 'lap_dnf')]


_hash_module = None
_hash_cache = {}            # key -> value; keys carry (path, mtime_ns, size) of every file the value was read from


def _load_source_hash_module():
    global _hash_module
    if _hash_module is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location('_mcgp_source_hash', os.path.join(CSRC, 'source_hash.py'))
        _hash_module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_hash_module)
    return _hash_module


def _sources():
    return _load_source_hash_module().sources()


def _stat_key(paths):
    key = []
    for path in sorted(paths):
        try:
            st = os.stat(path)
            key.append((path, st.st_mtime_ns, st.st_size))
        except OSError:
            key.append((path, None, None))
    return tuple(key)


def source_hash():
    """Identity of the kernel sources in this tree (csrc/source_hash.py): the Makefile compiles the same value into
    libmcgp_hip.so, and profiles under profiles/ are stamped with it.  Cached per process on (path, mtime, size) of
    every hashed file: every rank of a launch asks several times at start-up."""
    key = ('src',) + _stat_key(_sources())
    if key not in _hash_cache:
        for k in [k for k in _hash_cache if k[0] == 'src']:
            del _hash_cache[k]
        _hash_cache[key] = _load_source_hash_module().source_hash()
    return _hash_cache[key]


_MARKER = b'MCGP_BUILD_HASH='


def file_build_hash(path):
    """The source hash a library FILE was compiled from, read from its marker string without loading it
    (None: no such file, or a binary without the marker).  The file (tens of megabytes of kernels) is scanned in
    pieces, not read whole, and the answer is cached per process on (path, mtime, size)."""
    key = ('lib',) + _stat_key([path])
    if key in _hash_cache:
        return _hash_cache[key]
    found = None
    try:
        with open(path, 'rb') as f:
            tail = b''
            while found is None:
                piece = f.read(1 << 20)
                if not piece:
                    break
                blob = tail + piece
                i = blob.find(_MARKER)
                if i >= 0:
                    rest = blob[i + len(_MARKER):]
                    while b'\0' not in rest:                    # the value straddles the end of this piece
                        more = f.read(4096)
                        if not more:
                            break
                        rest += more
                    found = rest.split(b'\0', 1)[0].decode('ascii', 'replace')
                tail = blob[-(len(_MARKER) - 1):]
    except OSError:
        return None
    for k in [k for k in _hash_cache if k[0] == 'lib' and k[1][0] == path]:
        del _hash_cache[k]
    _hash_cache[key] = found
    return found


def _stale():
    """The library is stale unless it carries the hash of the sources beside it (contents, not time stamps: a stale
    binary with a newer mtime is still stale)."""
    return file_build_hash(LIB_PATH) != source_hash()


def build(force=False):
    """Compile csrc/ for gfx950 (hipcc cross-compiles without a GPU).

    One builder per node: an exclusive flock serialises the ranks of a torch.distributed.run launch (the
    first one in builds, the others find the library up to date), make relinks through a temporary file
    renamed into place, and an up-to-date tree is a no-op (make is incremental).  Runs before any HIP call of
    this process.  MCGP_NO_BUILD=1 forbids building (deployments without hipcc, and the staleness test): a stale
    library is then refused instead of rebuilt."""
    if not (force or _stale()):
        return LIB_PATH
    if os.environ.get('MCGP_NO_BUILD') == '1':
        raise McgpError(-1, f'{LIB_PATH} is stale (built from sources {file_build_hash(LIB_PATH)}, the tree is '
                            f'{source_hash()}) and MCGP_NO_BUILD=1 forbids rebuilding it')
    import fcntl
    os.makedirs(os.path.join(CSRC, 'build'), exist_ok=True)
    with open(os.path.join(CSRC, 'build', '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or _stale():
                subprocess.check_call(['make', '-C', CSRC, '-s'] + (['-B'] if force else []))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if _stale():
        raise McgpError(-1, f'{LIB_PATH} still carries source hash {file_build_hash(LIB_PATH)} after make; the tree '
                            f'is {source_hash()}')
    return LIB_PATH


def hip_runtimes_mapped():
    """Paths of the libamdhip64 copies mapped into this process (one in a healthy process)."""
    paths = set()
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                if 'libamdhip64' in line:
                    paths.add(line.split(None, 5)[-1].strip())
    except OSError:
        pass
    return sorted(paths)


def _bind_hip_runtime():
    """Make libmcgp_hip.so and PyTorch share ONE HIP runtime, whichever is loaded first.

    libmcgp_hip.so needs `libamdhip64.so.7` (found in /opt/rocm through its RUNPATH); PyTorch-ROCm ships its own
    copy and asks for it as `libamdhip64.so`, a name the loader does not match against an already mapped
    /opt/rocm copy.  Library first, torch second therefore maps two runtimes, and the second one finds no GPU
    ("No HIP GPUs are available"); a hipStream_t of one handed to the other is undefined behaviour.  So when a
    torch installation is present (located without importing it) and no runtime is mapped yet, map ITS copy
    globally first: the library's NEEDED entry then binds to it by SONAME, and a later `import torch` finds
    the same file.  MCGP_HIP_RUNTIME=system skips this (torch-free deployments), or names a libamdhip64 to use."""
    if hip_runtimes_mapped():
        return
    choice = os.environ.get('MCGP_HIP_RUNTIME', '')
    if choice == 'system':
        return
    cand = choice
    if not cand:
        import importlib.util
        try:
            spec = importlib.util.find_spec('torch')
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def assert_single_hip_runtime():
    """Raise if this process has ended up with two HIP runtimes (library loaded against one, torch.cuda on another)."""
    mapped = hip_runtimes_mapped()
    if len(mapped) > 1:
        raise McgpError(-3, 'two HIP runtimes are mapped into this process (' + ', '.join(mapped) + '): '
                        'streams and device state cannot be shared between them; import torch before '
                        'loading the library or set MCGP_HIP_RUNTIME')


_lib = None

EXPORTS = ('mcgp_abi_version', 'mcgp_build_hash', 'mcgp_run_batch', 'mcgp_device_count', 'mcgp_last_error', 'mcgp_run', 'mcgp_run_device',
           'mcgp_simulate_race', 'mcgp_grid_probs', 'mcgp_run_from_ratings', 'mcgp_last_kernel_ms',
           'mcgp_stream_kernel_ms', 'mcgp_elo_season',
           'mcgp_last_launch_info', 'mcgp_last_kernel_name')


def lib():
    """Load the library (building it if the sources are newer); raises if impossible."""
    global _lib
    if _lib is None:
        path = os.environ.get('MCGP_LIB')          # diagnostic builds (tools/ablate.sh)
        if not path:
            build()
            path = LIB_PATH
        _bind_hip_runtime()
        L = C.CDLL(path)
        assert_single_hip_runtime()
        if hasattr(L, 'mcgp_build_hash'):
            L.mcgp_build_hash.restype = C.c_char_p
        if not os.environ.get('MCGP_LIB'):
            # the binary that is MAPPED must be the one whose file was checked (a rebuild renames a new file into place)
            loaded = L.mcgp_build_hash().decode() if hasattr(L, 'mcgp_build_hash') else None
            if loaded != source_hash():
                raise McgpError(-1, f'{path} was compiled from sources {loaded}, the tree is {source_hash()}: refusing '
                                    'to run a stale kernel')
        L.mcgp_abi_version.restype = C.c_int32
        L.mcgp_device_count.restype = C.c_int32
        L.mcgp_last_error.restype = C.c_char_p
        dp = C.POINTER(C.c_double)
        L.mcgp_run.restype = C.c_int32
        L.mcgp_run.argtypes = [C.POINTER(McgpConfig), C.POINTER(McgpDrivers), dp, C.c_uint32, C.c_uint64,
                               C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)]
        if hasattr(L, 'mcgp_run_batch'):
            L.mcgp_run_batch.restype = C.c_int32
            L.mcgp_run_batch.argtypes = [C.c_uint32, C.POINTER(McgpConfig), C.POINTER(McgpDrivers), C.POINTER(dp), C.c_uint32,
                                         C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32,
                                         C.POINTER(C.c_uint64)]
        L.mcgp_run_device.restype = C.c_int32
        L.mcgp_run_device.argtypes = [C.POINTER(McgpConfig), C.POINTER(McgpDrivers), dp, C.c_uint32, C.c_uint64,
                                      C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcgp_simulate_race.restype = C.c_int32
        L.mcgp_simulate_race.argtypes = [C.POINTER(McgpConfig), C.POINTER(McgpDrivers), C.POINTER(C.c_uint8),
                                         C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_uint8)]
        ip = C.POINTER(C.c_int32)
        missing = [f for f in EXPORTS if not hasattr(L, f)]
        if missing and not os.environ.get('MCGP_LIB'):          # diagnostic builds of older sources may lack entry points
            raise McgpError(-1, f'{path} does not export {missing}')
        if 'mcgp_grid_probs' not in missing:
            L.mcgp_grid_probs.restype = C.c_int32
            L.mcgp_grid_probs.argtypes = [dp, dp, dp, dp, ip, C.c_uint32, C.c_int32, dp]
            L.mcgp_run_from_ratings.restype = C.c_int32
            L.mcgp_run_from_ratings.argtypes = [C.POINTER(McgpConfig), C.POINTER(McgpDrivers), dp, dp, dp, dp, ip,
                                                C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32,
                                                C.POINTER(C.c_uint64), dp]
        if 'mcgp_elo_season' not in missing:
            L.mcgp_elo_season.restype = C.c_int32
            L.mcgp_elo_season.argtypes = [C.c_uint32, C.c_uint32, ip, dp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), dp,
                                          dp, dp, C.c_int32]
        L.mcgp_last_kernel_ms.restype = C.c_int32
        L.mcgp_last_kernel_ms.argtypes = [C.c_int32, C.POINTER(C.c_float)]
        if 'mcgp_stream_kernel_ms' not in missing:
            L.mcgp_stream_kernel_ms.restype = C.c_int32
            L.mcgp_stream_kernel_ms.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_float)]
        L.mcgp_last_kernel_name.restype = C.c_char_p
        L.mcgp_last_kernel_name.argtypes = [C.c_int32]
        L.mcgp_last_launch_info.restype = C.c_int32
        L.mcgp_last_launch_info.argtypes = [C.c_int32] + [C.POINTER(C.c_uint32)] * 3
        if L.mcgp_abi_version() != ABI_VERSION:
            raise McgpError(-1, f'ABI version {L.mcgp_abi_version()} != {ABI_VERSION}')
        _lib = L
    return _lib


def build_hash():
    """Source hash compiled into the library this process has loaded (mcgp_build_hash())."""
    L = lib()
    return L.mcgp_build_hash().decode() if hasattr(L, 'mcgp_build_hash') else None


def check(rc):
    if rc != 0:
        raise McgpError(rc, lib().mcgp_last_error().decode('utf-8', 'replace'))
