"""CPU ORACLE (test infrastructure) -- the CONSUMER-side specification restated literally.

poseidon2_ref.py follows the producer side (the Haskell twin of the Nim tool).  This module restates the circom
templates the emitted input.json is fed to, signal for signal, so that tests can check that the two in-tree
specifications agree when executed (the reference never runs them against each other outside a full proof):

  circuit/poseidon2/poseidon2_perm.circom:10-198    SBox, InternalRound(i), ExternalRound(i), LinearLayer, Permutation
  circuit/poseidon2/poseidon2_sponge.circom:28-99   PoseidonSponge(t, capacity, input_len, output_len)
  circuit/poseidon2/poseidon2_hash.circom:12-31     Poseidon2_hash_rate1 / _rate2
  circuit/poseidon2/poseidon2_compr.circom:30-41    KeyedCompression

Constants: the circom file carries its own copy of the 80 round constants (internal: poseidon2_perm.circom:27-84,
external rows 0-3 initial / 4-7 final: :102-136); they were diffed against RoundConsts.hs (identical), so the
shared table oracle/p2_consts.py is indexed here the way the circom arrays are.
"""
from .p2_consts import ROUND_CONSTS

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # circom's bn128 field

INTERNAL = ROUND_CONSTS[12:68]                                   # round_consts[56], poseidon2_perm.circom:27-84
EXTERNAL = [ROUND_CONSTS[3 * i:3 * i + 3] for i in range(4)] + [ROUND_CONSTS[68 + 3 * i:68 + 3 * i + 3] for i in range(4)]   # [8][3]


def SBox(inp):                                                   # :10-18
    x2 = inp * inp % P
    x4 = x2 * x2 % P
    return inp * x4 % P


def InternalRound(i, inp):                                       # :23-92
    sb = SBox((inp[0] + INTERNAL[i]) % P)
    return [(2 * sb + inp[1] + inp[2]) % P, (sb + 2 * inp[1] + inp[2]) % P, (sb + inp[1] + 3 * inp[2]) % P]


def ExternalRound(i, inp):                                       # :97-148
    sb = [SBox((inp[j] + EXTERNAL[i][j]) % P) for j in range(3)]
    return [(2 * sb[0] + sb[1] + sb[2]) % P, (sb[0] + 2 * sb[1] + sb[2]) % P, (sb[0] + sb[1] + 2 * sb[2]) % P]


def LinearLayer(inp):                                            # :153-159
    return [(2 * inp[0] + inp[1] + inp[2]) % P, (inp[0] + 2 * inp[1] + inp[2]) % P, (inp[0] + inp[1] + 2 * inp[2]) % P]


def Permutation(inp):                                            # :164-198
    aux = [None] * 65
    aux[0] = LinearLayer([v % P for v in inp])
    for k in range(4):
        aux[k + 1] = ExternalRound(k, aux[k])
    for k in range(56):
        aux[k + 5] = InternalRound(k, aux[k + 4])
    for k in range(4):
        aux[k + 61] = ExternalRound(k + 4, aux[k + 60])
    return aux[64]


def PoseidonSponge(t, capacity, inp, output_len):                # poseidon2_sponge.circom:28-99
    rate = t - capacity
    assert t == 3 and 0 < capacity < t and 0 < rate < t
    input_len = len(inp)
    nblocks = ((input_len + 1) + (rate - 1)) // rate
    nout = (output_len + (rate - 1)) // rate
    padded_len = nblocks * rate
    padded = [v % P for v in inp] + [1] + [0] * (padded_len - input_len - 1)
    civ = 2 ** 64 + 256 * t + rate
    state = [0] * (t - 1) + [civ]
    for m in range(nblocks):
        sorbed = [(state[i] + padded[m * rate + i]) % P for i in range(rate)]
        state = Permutation(sorbed + state[rate:])
    out = state[:min(rate, output_len)]
    out_ptr = rate
    for n in range(1, nout):
        state = Permutation(state)
        out += state[:min(rate, output_len - out_ptr)]
        out_ptr += rate
    return out


def Poseidon2_hash_rate1(inp):                                   # poseidon2_hash.circom:12-19
    return PoseidonSponge(3, 2, inp, 1)[0]


def Poseidon2_hash_rate2(inp):                                   # poseidon2_hash.circom:24-31
    return PoseidonSponge(3, 1, inp, 1)[0]


def KeyedCompression(key, inp):                                  # poseidon2_compr.circom:30-41
    return Permutation([inp[0], inp[1], key])[0]
