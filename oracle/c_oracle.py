"""ctypes binding of the C oracle (oracle/p2_oracle.c).  TEST INFRASTRUCTURE ONLY -- see p2_oracle.h."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libp2oracle.so")
_lib = None


def build(force=False):
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("p2_oracle.c", "p2_oracle.h", "p2_consts.h"))
    if force or src_newer:
        # -march=native must not travel: the .so may run on a different host CPU than it was built on
        subprocess.check_call(["make", "-C", _HERE, "CFLAGS=-O3 -fPIC -Wall -Wextra -std=gnu11"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p, sz, u64, u32, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
        sigs = {
            "p2o_permute": (None, [u8p, u8p]),
            "p2o_permute_batch": (None, [u8p, u8p, sz]),
            "p2o_permute_batch_mt": (None, [u8p, u8p, sz, i32]),
            "p2o_compress": (None, [u8p, u8p, u32, u8p]),
            "p2o_sponge1_felts": (None, [u8p, sz, u8p]),
            "p2o_sponge2_felts": (None, [u8p, sz, u8p]),
            "p2o_hash_bytes": (None, [u8p, sz, u8p]),
            "p2o_felts_per_bytes": (sz, [sz]),
            "p2o_bytes_to_felts": (None, [u8p, sz, u8p]),
            "p2o_hash_cells": (None, [u8p, sz, sz, u8p]),
            "p2o_hash_cells_mt": (None, [u8p, sz, sz, u8p, i32]),
            "p2o_merkle_total": (sz, [sz]),
            "p2o_merkle_tree": (sz, [u8p, sz, u8p, ctypes.POINTER(ctypes.c_size_t)]),
            "p2o_merkle_root": (None, [u8p, sz, u8p]),
            "p2o_gen_fake_cell": (None, [u64, u64, sz, u8p]),
            "p2o_slot_seed": (u64, [u64, u64]),
            "p2o_fake_slot_root": (None, [u64, sz, sz, sz, u8p, i32]),
            "p2o_fake_slot_block_roots": (None, [u64, sz, sz, sz, u8p, i32]),
            "p2o_cell_index": (u64, [u8p, u8p, u64, u64]),
        }
        for name, (res, args) in sigs.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u8(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


# ---- integer <-> 32-byte LE helpers ---------------------------------------------------------
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


# ---- numpy-facing wrappers ------------------------------------------------------------------
def permute_batch(states, threads=1):
    s = _u8(states).reshape(-1, 96)
    out = np.empty_like(s)
    lib().p2o_permute_batch_mt(_ptr(s), _ptr(out), s.shape[0], threads)
    return out


def compress(x, y, key):
    x, y = _u8(x), _u8(y)
    out = np.empty(32, dtype=np.uint8)
    lib().p2o_compress(_ptr(x), _ptr(y), key, _ptr(out))
    return out


def sponge2_felts(felts):
    f = _u8(felts).reshape(-1, 32)
    out = np.empty(32, dtype=np.uint8)
    lib().p2o_sponge2_felts(_ptr(f), f.shape[0], _ptr(out))
    return out


def sponge1_felts(felts):
    f = _u8(felts).reshape(-1, 32)
    out = np.empty(32, dtype=np.uint8)
    lib().p2o_sponge1_felts(_ptr(f), f.shape[0], _ptr(out))
    return out


def hash_bytes(data):
    d = _u8(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else _u8(data)
    out = np.empty(32, dtype=np.uint8)
    lib().p2o_hash_bytes(_ptr(d) if d.size else None, d.size, _ptr(out))
    return out


def bytes_to_felts(data):
    d = _u8(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else _u8(data)
    n = lib().p2o_felts_per_bytes(d.size)
    out = np.empty((n, 32), dtype=np.uint8)
    lib().p2o_bytes_to_felts(_ptr(d) if d.size else None, d.size, _ptr(out))
    return out


def hash_cells(cells, cell_size, threads=1):
    c = _u8(cells).reshape(-1)
    n = c.size // cell_size
    out = np.empty((n, 32), dtype=np.uint8)
    lib().p2o_hash_cells_mt(_ptr(c), cell_size, n, _ptr(out), threads)
    return out


def merkle_tree(leaves):
    """Returns a list of layers (each an (m,32) uint8 array), bottom first."""
    lv = _u8(leaves).reshape(-1, 32)
    n = lv.shape[0]
    total = lib().p2o_merkle_total(n)
    out = np.empty((total, 32), dtype=np.uint8)
    sizes = (ctypes.c_size_t * 80)()
    nl = lib().p2o_merkle_tree(_ptr(lv), n, _ptr(out), sizes)
    layers, off = [], 0
    for i in range(nl):
        layers.append(out[off:off + sizes[i]])
        off += sizes[i]
    return layers


def merkle_root(leaves):
    lv = _u8(leaves).reshape(-1, 32)
    out = np.empty(32, dtype=np.uint8)
    lib().p2o_merkle_root(_ptr(lv), lv.shape[0], _ptr(out))
    return out


def gen_fake_cell(seed, idx, cell_size):
    out = np.empty(cell_size, dtype=np.uint8)
    lib().p2o_gen_fake_cell(seed, idx, cell_size, _ptr(out))
    return out


def gen_fake_cells(seed, first, n, cell_size):
    out = np.empty((n, cell_size), dtype=np.uint8)
    for i in range(n):
        lib().p2o_gen_fake_cell(seed, first + i, cell_size, ctypes.c_void_p(out[i].ctypes.data))
    return out


def slot_seed(seed, slot_idx):
    return lib().p2o_slot_seed(seed, slot_idx)


def fake_slot_root(slot_seed_, cell_size, block_size, n_cells, threads=1):
    out = np.empty(32, dtype=np.uint8)
    lib().p2o_fake_slot_root(slot_seed_, cell_size, block_size, n_cells, _ptr(out), threads)
    return out


def fake_slot_block_roots(slot_seed_, cell_size, block_size, n_cells, threads=1):
    """(n_cells / cellsPerBlock, 32) uint8: the root of every block tree of a fake-data slot."""
    out = np.empty((n_cells // (block_size // cell_size), 32), dtype=np.uint8)
    lib().p2o_fake_slot_block_roots(slot_seed_, cell_size, block_size, n_cells, _ptr(out), threads)
    return out


def cell_index(entropy, slot_root, n_cells, counter):
    e, s = _u8(entropy), _u8(slot_root)
    return lib().p2o_cell_index(_ptr(e), _ptr(s), n_cells, counter)
