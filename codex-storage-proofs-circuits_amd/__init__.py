"""ctypes binding of libcodex_p2.so (include/codex_p2.h) for tests and bench.py.

This is glue, not the product: the product is the HIP library and its C ABI, with the C++ host
layer (`host/`) mirroring the reference's Nim interface.  The binding never computes a hash itself and
has no CPU fallback -- if the library is missing or no gfx950 GPU is usable it raises.

The directory name contains '-', so import it through `__graft_entry__.load_package()`.
"""
import ctypes
import os
import subprocess
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION_MAJOR, ABI_VERSION_MINOR = 1, 0      # CP2_ABI_VERSION_* of the include/codex_p2.h this binding was written against (tests keep them equal)
LIB_PATH = os.environ.get("CODEX_P2_LIB") or os.path.join(_HERE, "libcodex_p2.so")   # env override: kernel-variant A/B runs
CLI_PATH = os.path.join(_HERE, "cli")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "codex_p2.h")

CP2_OK = 0
FELT = 32


def _free_children(owner):
    """Datasets and tree batches live inside their context (device buffers, streams, the scratch pool): whatever of them is
    still alive when the context / multi handle is closed is freed FIRST, so a late `free()` (or a destructor) finds nothing
    left to touch -- the C ABI's rule ("free every dataset made through a handle before the handle") kept by the binding."""
    for child in list(getattr(owner, "_children", ())):
        try:
            child.free()
        except Exception:
            pass


def _finalizing(_is_finalizing=sys.is_finalizing):
    """True at interpreter shutdown (module globals may already be None there, hence the bound default)."""
    try:
        return _is_finalizing()
    except Exception:
        return True


class CodexP2Error(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s failed: status %d%s" % (where, status, (" (" + detail + ")") if detail else ""))


class Config(ctypes.Structure):
    """cp2_config (GlobalConfig + DataSetConfig of reference/nim/proof_input/src/types.nim:82-101)."""
    _fields_ = [("max_depth", ctypes.c_int32), ("max_log2_nslots", ctypes.c_int32),
                ("cell_size", ctypes.c_uint64), ("block_size", ctypes.c_uint64),
                ("n_slots", ctypes.c_uint64), ("n_cells", ctypes.c_uint64),
                ("n_samples", ctypes.c_uint64), ("seed", ctypes.c_uint64),
                ("file_base", ctypes.c_char_p)]


def build(force=False, verbose=False):
    """Compile libcodex_p2.so and the cli twin in-tree with hipcc for gfx950."""
    args = ["make", "-C", _HERE] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen the in-tree library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64; two HIP runtimes in one process do not work
    # ("No HIP GPUs are available" from whichever comes second).  When torch is installed, load it first
    # so that this library binds to the runtime torch already mapped (same SONAME).
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    # The entry points below are bound by name: the boundary's version is what tells a library built from another header
    # (include/codex_p2.h, CP2_ABI_VERSION_*).  Another major, or a minor older than this binding's, is refused.
    try:
        L.cp2_abi_version.restype, L.cp2_abi_version.argtypes = ctypes.c_int, []
        v = L.cp2_abi_version()
    except AttributeError:
        v = None
    if v is None:
        if not os.environ.get("CODEX_P2_LIB"):          # (A/B tooling may name an older library on purpose)
            raise RuntimeError("%s predates the versioned boundary (no cp2_abi_version): rebuild it" % LIB_PATH)
    elif (v >> 16) != ABI_VERSION_MAJOR or (v & 0xffff) < ABI_VERSION_MINOR:
        raise RuntimeError("%s has ABI version %d.%d, this binding was written against %d.%d (include/codex_p2.h)" %
                           (LIB_PATH, v >> 16, v & 0xffff, ABI_VERSION_MAJOR, ABI_VERSION_MINOR))
    vp, sz, u64, u32, i32, cp = (ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                 ctypes.c_char_p)
    pvp = ctypes.POINTER(ctypes.c_void_p)
    sigs = {
        "cp2_init": (i32, [i32, pvp]),
        "cp2_free": (None, [vp]),
        "cp2_set_stream": (i32, [vp, vp]),
        "cp2_reset_stream": (i32, [vp]),
        "cp2_sync": (i32, [vp]),
        "cp2_strerror": (cp, [i32]),
        "cp2_last_error": (cp, [vp]),
        "cp2_device_is_native": (i32, [vp]),
        "cp2_check_environment": (i32, [cp, sz]),
        "cp2_abi_version": (i32, []),
        "cp2_set_ingest": (i32, [vp, i32, i32, sz]),
        "cp2_set_ingest_direct": (i32, [vp, i32]),
        "cp2_set_ingest_mapped": (i32, [vp, i32]),
        "cp2_trim": (i32, [vp]),
        "cp2_set_body_budget": (i32, [vp, sz, cp]),
        "cp2_set_keep_trees": (i32, [vp, i32]),
        "cp2_dataset_keeps_trees": (i32, [vp]),
        "cp2_permute_batch": (i32, [vp, vp, vp, sz]),
        "cp2_permute_batch_dev": (i32, [vp, vp, vp, sz]),
        "cp2_compress_batch": (i32, [vp, vp, u32, vp, sz]),
        "cp2_sponge2_felts": (i32, [vp, vp, sz, vp]),
        "cp2_sponge2_felts_batch": (i32, [vp, vp, sz, sz, vp]),
        "cp2_sponge2_felts_batch_dev": (i32, [vp, vp, sz, sz, vp]),
        "cp2_felts_per_bytes": (sz, [sz]),
        "cp2_bytes_to_felts": (i32, [vp, sz, vp]),
        "cp2_hash_cells": (i32, [vp, vp, sz, sz, vp]),
        "cp2_hash_cells_dev": (i32, [vp, vp, sz, sz, vp]),
        "cp2_hash_bytes": (i32, [vp, vp, sz, vp]),
        "cp2_merkle_total": (sz, [sz]),
        "cp2_merkle_num_layers": (sz, [sz]),
        "cp2_merkle_tree": (i32, [vp, vp, sz, vp, ctypes.POINTER(sz), ctypes.POINTER(sz)]),
        "cp2_merkle_trees_dev": (i32, [vp, vp, sz, sz, vp]),
        "cp2_merkle_root": (i32, [vp, vp, sz, vp]),
        "cp2_slot_seed": (u64, [u64, u64]),
        "cp2_gen_fake_cells": (i32, [vp, u64, u64, sz, sz, vp]),
        "cp2_gen_fake_cells_dev": (i32, [vp, u64, u64, sz, sz, vp]),
        "cp2_cell_indices": (i32, [vp, vp, vp, u64, sz, vp]),
        "cp2_slot_trees_build_fake": (i32, [vp, u64, u64, sz, sz, sz, sz, pvp]),
        "cp2_slot_trees_build_fake_units": (i32, [vp, u64, u64, u64, sz, sz, sz, sz, pvp]),
        "cp2_slot_trees_build_file_units": (i32, [vp, cp, u64, u64, sz, sz, sz, sz, pvp]),
        "cp2_slot_trees_build_dev": (i32, [vp, vp, sz, sz, sz, sz, pvp]),
        "cp2_slot_trees_build_host": (i32, [vp, vp, sz, sz, sz, sz, pvp]),
        "cp2_slot_trees_free": (None, [vp]),
        "cp2_slot_trees_save": (i32, [vp, cp]),
        "cp2_slot_trees_load": (i32, [vp, cp, pvp]),
        "cp2_slot_trees_attach_cells": (i32, [vp, vp, vp]),
        "cp2_dataset_build_cached": (i32, [vp, ctypes.POINTER(Config), u64, u64, cp, pvp]),
        "cp2_slot_trees_count": (sz, [vp]),
        "cp2_slot_trees_depth": (sz, [vp]),
        "cp2_slot_trees_roots": (i32, [vp, vp]),
        "cp2_slot_trees_roots_dev": (vp, [vp]),
        "cp2_slot_trees_paths": (i32, [vp, sz, vp, sz, sz, vp, vp]),
        "cp2_dataset_build": (i32, [vp, ctypes.POINTER(Config), u64, u64, pvp]),
        "cp2_dataset_free": (None, [vp]),
        "cp2_dataset_local_roots": (i32, [vp, vp]),
        "cp2_dataset_set_roots": (i32, [vp, vp]),
        "cp2_dataset_root": (i32, [vp, vp]),
        "cp2_dataset_local_roots_dev": (vp, [vp]),
        "cp2_dataset_copy_local_roots_dev": (i32, [vp, vp]),
        "cp2_dataset_set_roots_dev": (i32, [vp, vp]),
        "cp2_dataset_range": (i32, [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        "cp2_dataset_ctx": (vp, [vp]),
        "cp2_multi_init": (i32, [ctypes.POINTER(ctypes.c_int), i32, pvp]),
        "cp2_multi_free": (None, [vp]),
        "cp2_multi_count": (i32, [vp]),
        "cp2_multi_device": (i32, [vp, i32]),
        "cp2_multi_ctx": (vp, [vp, i32]),
        "cp2_multi_last_error": (cp, [vp]),
        "cp2_multi_gather_mode": (cp, [vp]),
        "cp2_multi_set_policy": (i32, [vp, i32, u64]),
        "cp2_multi_set_split": (i32, [vp, ctypes.c_int64]),
        "cp2_multi_dataset_units_per_slot": (u64, [vp]),
        "cp2_shard_range": (None, [u64, i32, i32, ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        "cp2_multi_plan": (i32, [ctypes.POINTER(Config), i32, u64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(u64)]),
        "cp2_multi_dataset_build": (i32, [vp, ctypes.POINTER(Config), pvp]),
        "cp2_multi_dataset_build_cached": (i32, [vp, ctypes.POINTER(Config), cp, pvp]),
        "cp2_multi_dataset_build_streamed": (i32, [vp, ctypes.POINTER(Config), vp, i32, sz, pvp]),
        "cp2_multi_dataset_free": (None, [vp]),
        "cp2_multi_dataset_shards": (i32, [vp]),
        "cp2_multi_dataset_shard": (vp, [vp, i32, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        "cp2_multi_dataset_root": (i32, [vp, vp]),
        "cp2_multi_dataset_slot_roots": (i32, [vp, vp]),
        "cp2_multi_proof_input_generate": (i32, [vp, u64, vp, pvp]),
        "cp2_multi_dataset_export_proof_inputs": (i32, [vp, vp, sz, vp, cp, i32, sz, ctypes.POINTER(u64)]),
        "cp2_multi_dataset_export_streamed": (i32, [vp, cp, i32, ctypes.POINTER(u64)]),
        "cp2_multi_dataset_streamed_json": (i32, [vp, u64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz)]),
        "cp2_proof_input_generate": (i32, [vp, u64, vp, pvp]),
        "cp2_proof_inputs_generate_batch": (i32, [vp, vp, sz, vp, pvp]),
        "cp2_proof_inputs_write_json_batch": (i32, [pvp, sz, ctypes.POINTER(cp), i32, ctypes.POINTER(u64)]),
        "cp2_dataset_export_proof_inputs": (i32, [vp, vp, sz, vp, cp, i32, sz, ctypes.POINTER(u64)]),
        "cp2_dataset_build_streamed": (i32, [vp, ctypes.POINTER(Config), u64, u64, vp, i32, sz, pvp]),
        "cp2_dataset_export_streamed": (i32, [vp, cp, i32, ctypes.POINTER(u64)]),
        "cp2_dataset_streamed_json": (i32, [vp, u64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz)]),
        "cp2_proof_input_free": (None, [vp]),
        "cp2_proof_input_roots": (i32, [vp, vp, vp, vp]),
        "cp2_proof_input_nsamples": (sz, [vp]),
        "cp2_proof_input_cell_indices": (vp, [vp]),
        "cp2_proof_input_cell_data": (vp, [vp]),
        "cp2_proof_input_merkle_paths": (vp, [vp]),
        "cp2_proof_input_slot_proof": (vp, [vp]),
        "cp2_proof_input_leaf_hashes": (vp, [vp]),
        "cp2_proof_input_create": (i32, [ctypes.POINTER(Config), u64, vp, vp, vp, vp, sz, vp, vp, vp, vp, pvp]),
        "cp2_proof_input_write_json": (i32, [vp, cp]),
        "cp2_proof_input_json": (i32, [vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz)]),
        "cp2_free_buffer": (None, [vp]),
        "cp2_write_circom_main": (i32, [ctypes.POINTER(Config), cp]),
    }
    for name, (res, args) in sigs.items():
        if v is None and name == "cp2_abi_version":
            continue           # (an older library named on purpose through CODEX_P2_LIB: A/B tooling)
        f = getattr(L, name)   # AttributeError if the header and the library ever disagree
        f.restype, f.argtypes = res, args
    L._cp2_signatures = sigs
    _lib = L
    return L


def check_environment():
    """cp2_check_environment: None when every CODEX_P2_* variable holds what it takes, else the message naming the first that does not."""
    L = load_library()
    buf = ctypes.create_string_buffer(512)
    return None if L.cp2_check_environment(buf, len(buf)) == CP2_OK else buf.value.decode()


def exported_symbols():
    """Names declared in include/codex_p2.h (parsed from the header text)."""
    import re
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cp2_[a-z0-9_]+)\s*\(", text)))


def _u8(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def felt_bytes(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint8).copy()


def felts_to_array(xs):
    out = np.zeros((len(xs), 32), dtype=np.uint8)
    for i, x in enumerate(xs):
        out[i] = felt_bytes(x)
    return out


def array_to_felts(a):
    a = _u8(a).reshape(-1, 32)
    return [int.from_bytes(a[i].tobytes(), "little") for i in range(a.shape[0])]


class Context:
    """One cp2_ctx.  Host-array methods mirror the host-pointer ABI; *_dev methods take raw device pointers."""

    def __init__(self, device=0):
        self.L = load_library()
        h = ctypes.c_void_p()
        st = self.L.cp2_init(device, ctypes.byref(h))
        if st != CP2_OK:
            raise CodexP2Error(st, "cp2_init", self.L.cp2_strerror(st).decode())
        self.h = h
        self._children = weakref.WeakSet()

    def close(self):
        if self.h:
            _free_children(self)
            self.L.cp2_free(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime (and this module's globals) may already be gone: leak instead
        try:
            if _finalizing():
                return
            self.close()
        except Exception:
            pass

    def _ck(self, st, where):
        if st != CP2_OK:
            raise CodexP2Error(st, where, self.L.cp2_last_error(self.h).decode() or self.L.cp2_strerror(st).decode())

    # -- plumbing
    def set_stream(self, stream_ptr):
        self._ck(self.L.cp2_set_stream(self.h, ctypes.c_void_p(stream_ptr)), "cp2_set_stream")

    def reset_stream(self):
        self._ck(self.L.cp2_reset_stream(self.h), "cp2_reset_stream")

    def sync(self):
        self._ck(self.L.cp2_sync(self.h), "cp2_sync")

    def set_ingest(self, fill_threads=0, ring_depth=0, chunk_bytes=0):
        self._ck(self.L.cp2_set_ingest(self.h, fill_threads, ring_depth, chunk_bytes), "cp2_set_ingest")

    def set_ingest_direct(self, on):
        """O_DIRECT reads of slot files (1 / 0; -1 = environment CP2_INGEST_DIRECT)."""
        self._ck(self.L.cp2_set_ingest_direct(self.h, on), "cp2_set_ingest_direct")

    def set_ingest_mapped(self, on):
        """page-cache-resident chunks of slot files uploaded straight from a mapping, no CPU copy (1 / 0; -1 = environment CP2_INGEST_MAPPED, default off)"""
        self._ck(self.L.cp2_set_ingest_mapped(self.h, on), "cp2_set_ingest_mapped")

    def trim(self):
        """Give the context's cached device / pinned scratch back to the system (cp2_trim)."""
        self._ck(self.L.cp2_trim(self.h), "cp2_trim")

    def set_keep_trees(self, mode=-1):
        """cp2_dataset_build keeps every node of the slot trees resident (1), the block roots and up (2: compact), only the
        roots (0), or the most that fits (-1)"""
        self._ck(self.L.cp2_set_keep_trees(self.h, mode), "cp2_set_keep_trees")

    def set_body_budget(self, max_resident_bytes=0, spill_dir=None):
        self._ck(self.L.cp2_set_body_budget(self.h, max_resident_bytes, spill_dir.encode() if spill_dir else None), "cp2_set_body_budget")

    # -- a1
    def permute_batch(self, states, out=None):
        s = _u8(states).reshape(-1, 96)
        if out is None:
            out = np.empty_like(s)
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size == s.size
        self._ck(self.L.cp2_permute_batch(self.h, _p(s), _p(out), s.shape[0]), "cp2_permute_batch")
        return out

    def permute_batch_dev(self, d_in, d_out, n):
        self._ck(self.L.cp2_permute_batch_dev(self.h, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), n), "cp2_permute_batch_dev")

    # -- a6
    def compress_batch(self, xy, key):
        a = _u8(xy).reshape(-1, 64)
        out = np.empty((a.shape[0], 32), dtype=np.uint8)
        self._ck(self.L.cp2_compress_batch(self.h, _p(a), key, _p(out), a.shape[0]), "cp2_compress_batch")
        return out

    # -- a3
    def sponge2_felts(self, felts):
        f = _u8(felts).reshape(-1, 32)
        out = np.empty(32, dtype=np.uint8)
        self._ck(self.L.cp2_sponge2_felts(self.h, _p(f) if f.size else None, f.shape[0], _p(out)), "cp2_sponge2_felts")
        return out

    def sponge2_felts_batch(self, felts, nf):
        f = _u8(felts).reshape(-1, 32)
        nitems = f.shape[0] // nf if nf else 0
        out = np.empty((nitems, 32), dtype=np.uint8)
        self._ck(self.L.cp2_sponge2_felts_batch(self.h, _p(f), nf, nitems, _p(out)), "cp2_sponge2_felts_batch")
        return out

    # -- a4
    def bytes_to_felts(self, data):
        d = _u8(np.frombuffer(bytes(data), dtype=np.uint8))
        n = self.L.cp2_felts_per_bytes(d.size)
        out = np.empty((n, 32), dtype=np.uint8)
        self._ck(self.L.cp2_bytes_to_felts(_p(d) if d.size else None, d.size, _p(out)), "cp2_bytes_to_felts")
        return out

    # -- a5
    def hash_cells(self, cells, cell_size):
        c = _u8(cells).reshape(-1)
        n = c.size // cell_size if cell_size else 0
        out = np.empty((n, 32), dtype=np.uint8)
        self._ck(self.L.cp2_hash_cells(self.h, _p(c), cell_size, n, _p(out)), "cp2_hash_cells")
        return out

    def hash_cells_dev(self, d_cells, cell_size, n_cells, d_out):
        self._ck(self.L.cp2_hash_cells_dev(self.h, ctypes.c_void_p(d_cells), cell_size, n_cells, ctypes.c_void_p(d_out)), "cp2_hash_cells_dev")

    def hash_bytes(self, data):
        d = _u8(np.frombuffer(bytes(data), dtype=np.uint8))
        out = np.empty(32, dtype=np.uint8)
        self._ck(self.L.cp2_hash_bytes(self.h, _p(d) if d.size else None, d.size, _p(out)), "cp2_hash_bytes")
        return out

    # -- a7
    def merkle_tree(self, leaves):
        lv = _u8(leaves).reshape(-1, 32)
        n = lv.shape[0]
        total = self.L.cp2_merkle_total(n)
        out = np.empty((total, 32), dtype=np.uint8)
        sizes = (ctypes.c_size_t * 80)()
        nl = ctypes.c_size_t()
        self._ck(self.L.cp2_merkle_tree(self.h, _p(lv) if n else None, n, _p(out), sizes, ctypes.byref(nl)), "cp2_merkle_tree")
        layers, off = [], 0
        for i in range(nl.value):
            layers.append(out[off:off + sizes[i]])
            off += sizes[i]
        return layers

    def merkle_trees_dev(self, d_leaves, n, nseg, d_layers):
        self._ck(self.L.cp2_merkle_trees_dev(self.h, ctypes.c_void_p(d_leaves), n, nseg, ctypes.c_void_p(d_layers)), "cp2_merkle_trees_dev")

    def merkle_root(self, leaves):
        lv = _u8(leaves).reshape(-1, 32)
        out = np.empty(32, dtype=np.uint8)
        self._ck(self.L.cp2_merkle_root(self.h, _p(lv) if lv.size else None, lv.shape[0], _p(out)), "cp2_merkle_root")
        return out

    # -- a10
    def slot_seed(self, seed, slot_idx):
        return self.L.cp2_slot_seed(seed, slot_idx)

    def gen_fake_cells(self, seed, first, n, cell_size):
        out = np.empty((n, cell_size), dtype=np.uint8)
        self._ck(self.L.cp2_gen_fake_cells(self.h, seed, first, n, cell_size, _p(out)), "cp2_gen_fake_cells")
        return out

    def gen_fake_cells_dev(self, seed, first, n, cell_size, d_out):
        self._ck(self.L.cp2_gen_fake_cells_dev(self.h, seed, first, n, cell_size, ctypes.c_void_p(d_out)), "cp2_gen_fake_cells_dev")

    # -- a12
    def cell_indices(self, entropy, slot_root, n_cells, n_samples):
        e, r = _u8(entropy), _u8(slot_root)
        out = np.empty(n_samples, dtype=np.uint64)
        self._ck(self.L.cp2_cell_indices(self.h, _p(e), _p(r), n_cells, n_samples, _p(out)), "cp2_cell_indices")
        return out

    # -- slot trees
    def slot_trees_fake(self, dataset_seed, first_slot, n_slots, cell_size, block_size, n_cells):
        h = ctypes.c_void_p()
        self._ck(self.L.cp2_slot_trees_build_fake(self.h, dataset_seed, first_slot, n_slots, cell_size, block_size, n_cells,
                                                  ctypes.byref(h)), "cp2_slot_trees_build_fake")
        return SlotTrees(self, h)

    def slot_trees_fake_units(self, dataset_seed, units_per_slot, first_unit, n_units, cell_size, block_size, cells_per_unit):
        h = ctypes.c_void_p()
        self._ck(self.L.cp2_slot_trees_build_fake_units(self.h, dataset_seed, units_per_slot, first_unit, n_units, cell_size, block_size,
                                                        cells_per_unit, ctypes.byref(h)), "cp2_slot_trees_build_fake_units")
        return SlotTrees(self, h)

    def slot_trees_file_units(self, file_base, units_per_slot, first_unit, n_units, cell_size, block_size, cells_per_unit):
        h = ctypes.c_void_p()
        self._ck(self.L.cp2_slot_trees_build_file_units(self.h, file_base.encode(), units_per_slot, first_unit, n_units, cell_size, block_size,
                                                        cells_per_unit, ctypes.byref(h)), "cp2_slot_trees_build_file_units")
        return SlotTrees(self, h)

    def slot_trees_dev(self, d_cells, n_slots, cell_size, block_size, n_cells):
        h = ctypes.c_void_p()
        self._ck(self.L.cp2_slot_trees_build_dev(self.h, ctypes.c_void_p(d_cells), n_slots, cell_size, block_size, n_cells,
                                                 ctypes.byref(h)), "cp2_slot_trees_build_dev")
        return SlotTrees(self, h)

    def slot_trees_host(self, cells, n_slots, cell_size, block_size, n_cells):
        c = _u8(cells).reshape(-1)
        h = ctypes.c_void_p()
        self._ck(self.L.cp2_slot_trees_build_host(self.h, _p(c), n_slots, cell_size, block_size, n_cells, ctypes.byref(h)),
                 "cp2_slot_trees_build_host")
        t = SlotTrees(self, h)
        t._keep = c   # the library keeps the host pointer for sampled-cell retrieval
        return t

    def slot_trees_load(self, path):
        h = ctypes.c_void_p()
        self._ck(self.L.cp2_slot_trees_load(self.h, path.encode(), ctypes.byref(h)), "cp2_slot_trees_load")
        return SlotTrees(self, h)

    # -- dataset / proof input
    def dataset(self, cfg, first_slot=0, n_local=None, cache=None):
        return Dataset(self, cfg, first_slot, cfg.n_slots if n_local is None else n_local, cache)

    def dataset_streamed(self, cfg, entropy, first_slot=0, n_local=None, threads=1, group_slots=0):
        """cp2_dataset_build_streamed: trees + (overlapped) the proof-input bodies of every local slot for `entropy`."""
        return Dataset(self, cfg, first_slot, cfg.n_slots if n_local is None else n_local, None,
                       streamed=(entropy, threads, group_slots))


def make_config(maxDepth=32, maxLog2NSlots=8, cellSize=2048, blockSize=65536, nSlots=11, nCells=256, nSamples=5,
                seed=12345, file=None):
    """Defaults are the reference CLI's (reference/nim/proof_input/src/cli.nim:47-76)."""
    c = Config()
    c.max_depth, c.max_log2_nslots = maxDepth, maxLog2NSlots
    c.cell_size, c.block_size = cellSize, blockSize
    c.n_slots, c.n_cells, c.n_samples, c.seed = nSlots, nCells, nSamples, seed
    c.file_base = file.encode() if file else None
    return c


class SlotTrees:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        ctx._children.add(self)

    def free(self):
        if self.h:
            self.ctx.L.cp2_slot_trees_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if _finalizing():
                return
            self.free()
        except Exception:
            pass

    @property
    def count(self):
        return self.ctx.L.cp2_slot_trees_count(self.h)

    @property
    def depth(self):
        return self.ctx.L.cp2_slot_trees_depth(self.h)

    def save(self, path):
        self.ctx._ck(self.ctx.L.cp2_slot_trees_save(self.h, path.encode()), "cp2_slot_trees_save")

    def roots(self):
        out = np.empty((self.count, 32), dtype=np.uint8)
        self.ctx._ck(self.ctx.L.cp2_slot_trees_roots(self.h, _p(out)), "cp2_slot_trees_roots")
        return out

    def roots_dev(self):
        return self.ctx.L.cp2_slot_trees_roots_dev(self.h)

    def paths(self, slot, cell_idx, max_depth):
        idx = np.ascontiguousarray(np.asarray(cell_idx, dtype=np.uint64))
        out = np.empty((idx.size, max_depth, 32), dtype=np.uint8)
        leaves = np.empty((idx.size, 32), dtype=np.uint8)
        self.ctx._ck(self.ctx.L.cp2_slot_trees_paths(self.h, slot, _p(idx), idx.size, max_depth, _p(out), _p(leaves)),
                     "cp2_slot_trees_paths")
        return out, leaves


class Dataset:
    def __init__(self, ctx, cfg, first_slot, n_local, cache=None, streamed=None):
        self.ctx, self.cfg = ctx, cfg
        h = ctypes.c_void_p()
        if streamed is not None:
            entropy, threads, group = streamed
            e = _u8(entropy if not isinstance(entropy, int) else felt_bytes(entropy))
            ctx._ck(ctx.L.cp2_dataset_build_streamed(ctx.h, ctypes.byref(cfg), first_slot, n_local, _p(e), threads, group,
                                                     ctypes.byref(h)), "cp2_dataset_build_streamed")
        elif cache:
            ctx._ck(ctx.L.cp2_dataset_build_cached(ctx.h, ctypes.byref(cfg), first_slot, n_local, cache.encode(), ctypes.byref(h)),
                    "cp2_dataset_build_cached")
        else:
            ctx._ck(ctx.L.cp2_dataset_build(ctx.h, ctypes.byref(cfg), first_slot, n_local, ctypes.byref(h)), "cp2_dataset_build")
        self.h, self.first_slot, self.n_local = h, first_slot, n_local
        ctx._children.add(self)

    def free(self):
        if self.h:
            self.ctx.L.cp2_dataset_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if _finalizing():
                return
            self.free()
        except Exception:
            pass

    def local_roots(self):
        out = np.empty((self.n_local, 32), dtype=np.uint8)
        self.ctx._ck(self.ctx.L.cp2_dataset_local_roots(self.h, _p(out)), "cp2_dataset_local_roots")
        return out

    def set_roots(self, all_roots=None):
        if all_roots is None:
            self.ctx._ck(self.ctx.L.cp2_dataset_set_roots(self.h, None), "cp2_dataset_set_roots")
        else:
            r = _u8(all_roots).reshape(-1, 32)
            assert r.shape[0] == self.cfg.n_slots
            self.ctx._ck(self.ctx.L.cp2_dataset_set_roots(self.h, _p(r)), "cp2_dataset_set_roots")

    @property
    def tree_mode(self):
        """1 every node resident, 2 compact (block roots and up), 0 roots only"""
        return self.ctx.L.cp2_dataset_keeps_trees(self.h)

    @property
    def keeps_trees(self):
        return self.tree_mode == 1

    def local_roots_dev(self):
        """device pointer to the n_local x 32 bytes of local slot roots"""
        return self.ctx.L.cp2_dataset_local_roots_dev(self.h)

    def copy_local_roots_dev(self, d_out):
        """enqueue a device-to-device copy of the local roots into the caller's buffer (context's stream; ctx.sync() after)"""
        self.ctx._ck(self.ctx.L.cp2_dataset_copy_local_roots_dev(self.h, ctypes.c_void_p(d_out)), "cp2_dataset_copy_local_roots_dev")

    def set_roots_dev(self, d_all_roots):
        """all n_slots roots from device memory (the gathered buffer): no host copy on the way in"""
        self.ctx._ck(self.ctx.L.cp2_dataset_set_roots_dev(self.h, ctypes.c_void_p(d_all_roots)), "cp2_dataset_set_roots_dev")

    def root(self):
        out = np.empty(32, dtype=np.uint8)
        self.ctx._ck(self.ctx.L.cp2_dataset_root(self.h, _p(out)), "cp2_dataset_root")
        return out

    def proof_input(self, slot_idx, entropy):
        e = _u8(entropy if not isinstance(entropy, int) else felt_bytes(entropy))
        h = ctypes.c_void_p()
        self.ctx._ck(self.ctx.L.cp2_proof_input_generate(self.h, slot_idx, _p(e), ctypes.byref(h)), "cp2_proof_input_generate")
        pi = ProofInput(self.ctx, h, self.cfg)
        pi.slot_idx = slot_idx
        return pi


    def export_proof_inputs(self, slot_indices, entropy, directory=None, threads=1, batch=0):
        """Pipelined generate + serialise (+ write "<directory>/input_<slot>.json"); returns the total text bytes."""
        e = _u8(entropy if not isinstance(entropy, int) else felt_bytes(entropy))
        idx = np.ascontiguousarray(np.asarray(slot_indices, dtype=np.uint64))
        total = ctypes.c_uint64()
        self.ctx._ck(self.ctx.L.cp2_dataset_export_proof_inputs(self.h, _p(idx), idx.size, _p(e), directory.encode() if directory else None,
                                                                threads, batch, ctypes.byref(total)), "cp2_dataset_export_proof_inputs")
        return total.value

    def export_streamed(self, directory=None, threads=1):
        """Finish the proof inputs prepared by Context.dataset_streamed; returns the total text bytes."""
        total = ctypes.c_uint64()
        self.ctx._ck(self.ctx.L.cp2_dataset_export_streamed(self.h, directory.encode() if directory else None, threads,
                                                            ctypes.byref(total)), "cp2_dataset_export_streamed")
        return total.value

    def streamed_json(self, slot_idx):
        text, ln = ctypes.c_void_p(), ctypes.c_size_t()
        self.ctx._ck(self.ctx.L.cp2_dataset_streamed_json(self.h, slot_idx, ctypes.byref(text), ctypes.byref(ln)), "cp2_dataset_streamed_json")
        s = ctypes.string_at(text, ln.value).decode()
        self.ctx.L.cp2_free_buffer(text)
        return s

    def proof_inputs(self, slot_indices, entropy):
        """Batched generateProofInput for many slots of this dataset (one sampling / gather / fetch)."""
        e = _u8(entropy if not isinstance(entropy, int) else felt_bytes(entropy))
        idx = np.ascontiguousarray(np.asarray(slot_indices, dtype=np.uint64))
        hs = (ctypes.c_void_p * idx.size)()
        self.ctx._ck(self.ctx.L.cp2_proof_inputs_generate_batch(self.h, _p(idx), idx.size, _p(e), hs), "cp2_proof_inputs_generate_batch")
        return [ProofInput(self.ctx, ctypes.c_void_p(h), self.cfg) for h in hs]


GATHER_AUTO, GATHER_RCCL, GATHER_HOST, GATHER_COPY = 0, 1, 2, 3


def shard_range(n_items, rank, world):
    """cp2_shard_range: (first, count) of the contiguous range rank `rank` of `world` holds."""
    L = load_library()
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    L.cp2_shard_range(n_items, rank, world, ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def multi_plan(cfg, n_devices, min_cells_per_device=0, units_per_slot=0):
    """cp2_multi_plan: (number of shards, units per slot) cp2_multi_dataset_build would use -- host-only arithmetic."""
    L = load_library()
    w, u = ctypes.c_int(), ctypes.c_uint64()
    st = L.cp2_multi_plan(ctypes.byref(cfg), n_devices, min_cells_per_device, units_per_slot, ctypes.byref(w), ctypes.byref(u))
    if st != CP2_OK:
        raise CodexP2Error(st, "cp2_multi_plan", L.cp2_strerror(st).decode())
    return w.value, u.value


class _BorrowedContext(Context):
    """A cp2_ctx owned by a cp2_multi (never freed from here)."""

    def __init__(self, L, h):
        self.L, self.h = L, h
        self._children = weakref.WeakSet()

    def close(self):
        _free_children(self)
        self.h = None


class Multi:
    """cp2_multi: one handle over several devices of the node (in-process sharding, include/codex_p2.h section e)."""

    def __init__(self, devices=None):
        self.L = load_library()
        h = ctypes.c_void_p()
        if devices:
            arr = (ctypes.c_int * len(devices))(*devices)
            st = self.L.cp2_multi_init(arr, len(devices), ctypes.byref(h))
        else:
            st = self.L.cp2_multi_init(None, 0, ctypes.byref(h))
        if st != CP2_OK:
            raise CodexP2Error(st, "cp2_multi_init", self.L.cp2_strerror(st).decode())
        self.h = h
        self._children = weakref.WeakSet()
        self._borrowed = {}

    def close(self):
        if self.h:
            _free_children(self)
            for c in self._borrowed.values():
                c.close()
            self._borrowed = {}
            self.L.cp2_multi_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if _finalizing():
                return
            self.close()
        except Exception:
            pass

    def _ck(self, st, where):
        if st != CP2_OK:
            raise CodexP2Error(st, where, self.L.cp2_multi_last_error(self.h).decode() or self.L.cp2_strerror(st).decode())

    @property
    def count(self):
        return self.L.cp2_multi_count(self.h)

    def devices(self):
        return [self.L.cp2_multi_device(self.h, i) for i in range(self.count)]

    def ctx(self, i=0):
        if i not in self._borrowed:
            h = self.L.cp2_multi_ctx(self.h, i)
            if not h:
                raise CodexP2Error(-2, "cp2_multi_ctx")
            self._borrowed[i] = _BorrowedContext(self.L, ctypes.c_void_p(h))   # one object per context: it tracks what was made through it
        return self._borrowed[i]

    def gather_mode(self):
        return self.L.cp2_multi_gather_mode(self.h).decode()

    def set_policy(self, gather=GATHER_AUTO, min_cells_per_device=0):
        self._ck(self.L.cp2_multi_set_policy(self.h, gather, min_cells_per_device), "cp2_multi_set_policy")

    def set_split(self, units_per_slot=0):
        """units every slot is cut into by Multi.dataset: 0 = choose, 1 = whole slots only, 2^k = exactly that"""
        self._ck(self.L.cp2_multi_set_split(self.h, units_per_slot), "cp2_multi_set_split")

    def dataset(self, cfg, cache=None):
        return MultiDataset(self, cfg, cache=cache)

    def dataset_streamed(self, cfg, entropy, threads=1, group_slots=0):
        return MultiDataset(self, cfg, streamed=(entropy, threads, group_slots))


class MultiDataset:
    def __init__(self, multi, cfg, cache=None, streamed=None):
        self.multi, self.cfg = multi, cfg
        L = multi.L
        h = ctypes.c_void_p()
        if streamed is not None:
            entropy, threads, group = streamed
            e = _u8(entropy if not isinstance(entropy, int) else felt_bytes(entropy))
            multi._ck(L.cp2_multi_dataset_build_streamed(multi.h, ctypes.byref(cfg), _p(e), threads, group, ctypes.byref(h)),
                      "cp2_multi_dataset_build_streamed")
        elif cache:
            multi._ck(L.cp2_multi_dataset_build_cached(multi.h, ctypes.byref(cfg), cache.encode(), ctypes.byref(h)), "cp2_multi_dataset_build_cached")
        else:
            multi._ck(L.cp2_multi_dataset_build(multi.h, ctypes.byref(cfg), ctypes.byref(h)), "cp2_multi_dataset_build")
        self.h = h
        multi._children.add(self)

    def free(self):
        if self.h:
            self.multi.L.cp2_multi_dataset_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if _finalizing():
                return
            self.free()
        except Exception:
            pass

    @property
    def units_per_slot(self):
        return self.multi.L.cp2_multi_dataset_units_per_slot(self.h)

    def shards(self):
        """[(device, first, count)] of every shard: slots, or units when units_per_slot > 1"""
        L, out = self.multi.L, []
        for i in range(L.cp2_multi_dataset_shards(self.h)):
            d, a, b = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64()
            L.cp2_multi_dataset_shard(self.h, i, ctypes.byref(d), ctypes.byref(a), ctypes.byref(b))
            out.append((d.value, a.value, b.value))
        return out

    def root(self):
        out = np.empty(32, dtype=np.uint8)
        self.multi._ck(self.multi.L.cp2_multi_dataset_root(self.h, _p(out)), "cp2_multi_dataset_root")
        return out

    def shard_root(self, i):
        """the dataset root as shard i's device computed it (every device builds the tree itself)"""
        L = self.multi.L
        ds = L.cp2_multi_dataset_shard(self.h, i, None, None, None)
        out = np.empty(32, dtype=np.uint8)
        self.multi._ck(L.cp2_dataset_root(ctypes.c_void_p(ds), _p(out)), "cp2_dataset_root")
        return out

    def slot_roots(self):
        out = np.empty((self.cfg.n_slots, 32), dtype=np.uint8)
        self.multi._ck(self.multi.L.cp2_multi_dataset_slot_roots(self.h, _p(out)), "cp2_multi_dataset_slot_roots")
        return out

    def proof_input(self, slot_idx, entropy):
        e = _u8(entropy if not isinstance(entropy, int) else felt_bytes(entropy))
        h = ctypes.c_void_p()
        self.multi._ck(self.multi.L.cp2_multi_proof_input_generate(self.h, slot_idx, _p(e), ctypes.byref(h)), "cp2_multi_proof_input_generate")
        pi = ProofInput(self.multi, h, self.cfg)
        pi.slot_idx = slot_idx
        return pi

    def export_proof_inputs(self, slot_indices, entropy, directory=None, threads=1, batch=0):
        e = _u8(entropy if not isinstance(entropy, int) else felt_bytes(entropy))
        idx = np.ascontiguousarray(np.asarray(slot_indices, dtype=np.uint64))
        total = ctypes.c_uint64()
        self.multi._ck(self.multi.L.cp2_multi_dataset_export_proof_inputs(self.h, _p(idx), idx.size, _p(e), directory.encode() if directory else None,
                                                                          threads, batch, ctypes.byref(total)), "cp2_multi_dataset_export_proof_inputs")
        return total.value

    def export_streamed(self, directory=None, threads=1):
        total = ctypes.c_uint64()
        self.multi._ck(self.multi.L.cp2_multi_dataset_export_streamed(self.h, directory.encode() if directory else None, threads,
                                                                      ctypes.byref(total)), "cp2_multi_dataset_export_streamed")
        return total.value

    def streamed_json(self, slot_idx):
        text, ln = ctypes.c_void_p(), ctypes.c_size_t()
        self.multi._ck(self.multi.L.cp2_multi_dataset_streamed_json(self.h, slot_idx, ctypes.byref(text), ctypes.byref(ln)), "cp2_multi_dataset_streamed_json")
        s = ctypes.string_at(text, ln.value).decode()
        self.multi.L.cp2_free_buffer(text)
        return s


def write_json_batch(ctx, proof_inputs, paths=None, threads=1):
    """Serialise (and write, when paths are given) many proof inputs on host threads; returns total text bytes."""
    n = len(proof_inputs)
    hs = (ctypes.c_void_p * n)(*[p.h for p in proof_inputs])
    ps = None
    if paths is not None:
        ps = (ctypes.c_char_p * n)(*[(q.encode() if q else None) for q in paths])
    total = ctypes.c_uint64()
    ctx._ck(ctx.L.cp2_proof_inputs_write_json_batch(hs, n, ps, threads, ctypes.byref(total)), "cp2_proof_inputs_write_json_batch")
    return total.value


class ProofInput:
    def __init__(self, ctx, h, cfg):
        self.ctx, self.h, self.cfg = ctx, h, cfg

    def free(self):
        if self.h:
            self.ctx.L.cp2_proof_input_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if _finalizing():
                return
            self.free()
        except Exception:
            pass

    def _arr(self, ptr, shape, dtype=np.uint8):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if n == 0:
            return np.zeros(shape, dtype=dtype)
        buf = (ctypes.c_uint8 * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

    def roots(self):
        d, s, e = (np.empty(32, dtype=np.uint8) for _ in range(3))
        self.ctx._ck(self.ctx.L.cp2_proof_input_roots(self.h, _p(d), _p(s), _p(e)), "cp2_proof_input_roots")
        return d, s, e

    def cell_indices(self):
        n = self.ctx.L.cp2_proof_input_nsamples(self.h)
        return self._arr(self.ctx.L.cp2_proof_input_cell_indices(self.h), (n,), np.uint64)

    def cell_data(self):
        n = self.ctx.L.cp2_proof_input_nsamples(self.h)
        return self._arr(self.ctx.L.cp2_proof_input_cell_data(self.h), (n, self.cfg.cell_size))

    def merkle_paths(self):
        n = self.ctx.L.cp2_proof_input_nsamples(self.h)
        return self._arr(self.ctx.L.cp2_proof_input_merkle_paths(self.h), (n, self.cfg.max_depth, 32))

    def slot_proof(self):
        return self._arr(self.ctx.L.cp2_proof_input_slot_proof(self.h), (self.cfg.max_log2_nslots, 32))

    def leaf_hashes(self):
        n = self.ctx.L.cp2_proof_input_nsamples(self.h)
        return self._arr(self.ctx.L.cp2_proof_input_leaf_hashes(self.h), (n, 32))

    def recreate(self):
        """A copy made through cp2_proof_input_create from this object's accessor arrays (what the Nim shim's
        exportProofInputBN254 does with a SlotProofInput value)."""
        d, s, e = self.roots()
        sp, idx, cells, paths, leaves = (np.ascontiguousarray(a) for a in
                                         (self.slot_proof(), self.cell_indices(), self.cell_data(), self.merkle_paths(), self.leaf_hashes()))
        slot_idx = getattr(self, "slot_idx", 0)
        h = ctypes.c_void_p()
        self.ctx._ck(self.ctx.L.cp2_proof_input_create(ctypes.byref(self.cfg), slot_idx, _p(d), _p(e), _p(s), _p(sp), idx.size,
                                                       _p(idx), _p(cells), _p(paths), _p(leaves), ctypes.byref(h)), "cp2_proof_input_create")
        q = ProofInput(self.ctx, h, self.cfg)
        q.slot_idx = slot_idx
        return q

    def json(self):
        text, ln = ctypes.c_void_p(), ctypes.c_size_t()
        self.ctx._ck(self.ctx.L.cp2_proof_input_json(self.h, ctypes.byref(text), ctypes.byref(ln)), "cp2_proof_input_json")
        s = ctypes.string_at(text, ln.value).decode()
        self.ctx.L.cp2_free_buffer(text)
        return s

    def write_json(self, path):
        self.ctx._ck(self.ctx.L.cp2_proof_input_write_json(self.h, path.encode()), "cp2_proof_input_write_json")


def write_circom_main(cfg, path):
    L = load_library()
    st = L.cp2_write_circom_main(ctypes.byref(cfg), path.encode())
    if st != CP2_OK:
        raise CodexP2Error(st, "cp2_write_circom_main", L.cp2_strerror(st).decode())
