## codex_p2.nim -- Nim binding of libcodex_p2.so (include/codex_p2.h).
##
## Drop-in replacement of the nim-poseidon2 / constantine call sites used by
## reference/nim/proof_input/src for --field=bn254 --hash=poseidon2 (SURVEY.md section 8b).
## NOTE: written against the C ABI by hand; there is no Nim compiler in the build image, so this file has
## never been compiled.  It is deliberately thin and mechanical: one `importc` per C entry point, plus the
## handful of procs whose names the reference's modules import.
##
## Usage inside the reference tree (INTEGRATION.md section 2): copy this file to reference/nim/proof_input/src/ and
## copy nim/overlay/src/* over the same-named modules there.  The overlay modules export the reference's own
## procs with the reference's own signatures (generateProofInputBN254, exportProofInputBN254, merkleTreeBN254,
## hashCell, cellIndices, ...), so cli.nim and every other importer compile unchanged; build with
##   nim c -d:release --passL:"-L<repo>/codex-storage-proofs-circuits_amd -lcodex_p2" src/cli.nim

import std/os                  # getEnv: CODEX_P2_CACHE, as in the cli twin

const libName = "libcodex_p2.so"
const
  abiVersionMajor* = 1         ## CP2_ABI_VERSION_MAJOR / _MINOR of the include/codex_p2.h this binding was written against
  abiVersionMinor* = 0         ## (tests/test_nim_binding.py keeps the two files equal)

type
  F* = array[32, byte]          ## canonical little-endian field element (NOT constantine's Montgomery limbs)
  Cp2Ctx = distinct pointer
  Cp2Dataset = distinct pointer
  Cp2ProofInput = distinct pointer
  Cp2Multi = distinct pointer          ## every GPU of the node behind one handle (include/codex_p2.h section e)
  Cp2MultiDataset = distinct pointer
  Cp2Config* {.bycopy.} = object
    maxDepth*, maxLog2NSlots*: int32
    cellSize*, blockSize*, nSlots*, nCells*, nSamples*, seed*: uint64
    fileBase*: cstring

{.push cdecl, dynlib: libName.}
proc cp2_abi_version(): cint {.importc.}
proc cp2_init(device: cint, ctx: ptr Cp2Ctx): cint {.importc.}
proc cp2_free(ctx: Cp2Ctx) {.importc.}
proc cp2_strerror(status: cint): cstring {.importc.}
proc cp2_last_error(ctx: Cp2Ctx): cstring {.importc.}
proc cp2_trim(ctx: Cp2Ctx): cint {.importc.}
proc cp2_set_ingest_direct(ctx: Cp2Ctx, on: cint): cint {.importc.}
proc cp2_set_ingest_mapped(ctx: Cp2Ctx, on: cint): cint {.importc.}
proc cp2_set_body_budget(ctx: Cp2Ctx, maxResidentBytes: csize_t, spillDir: cstring): cint {.importc.}
proc cp2_set_keep_trees(ctx: Cp2Ctx, mode: cint): cint {.importc.}
proc cp2_permute_batch(ctx: Cp2Ctx, inp, outp: ptr byte, n: csize_t): cint {.importc.}
proc cp2_compress_batch(ctx: Cp2Ctx, xy: ptr byte, key: uint32, outp: ptr byte, n: csize_t): cint {.importc.}
proc cp2_sponge2_felts(ctx: Cp2Ctx, felts: ptr byte, n: csize_t, outp: ptr byte): cint {.importc.}
proc cp2_felts_per_bytes(len: csize_t): csize_t {.importc.}
proc cp2_bytes_to_felts(data: ptr byte, len: csize_t, outp: ptr byte): cint {.importc.}
proc cp2_hash_cells(ctx: Cp2Ctx, cells: ptr byte, cellSize, nCells: csize_t, outp: ptr byte): cint {.importc.}
proc cp2_merkle_total(n: csize_t): csize_t {.importc.}
proc cp2_merkle_num_layers(n: csize_t): csize_t {.importc.}
proc cp2_merkle_tree(ctx: Cp2Ctx, leaves: ptr byte, n: csize_t, layersOut: ptr byte,
                     layerSizes: ptr csize_t, nLayers: ptr csize_t): cint {.importc.}
proc cp2_merkle_root(ctx: Cp2Ctx, leaves: ptr byte, n: csize_t, outp: ptr byte): cint {.importc.}
proc cp2_gen_fake_cells(ctx: Cp2Ctx, seed, first: uint64, n, cellSize: csize_t, outp: ptr byte): cint {.importc.}
proc cp2_cell_indices(ctx: Cp2Ctx, entropy, slotRoot: ptr byte, nCells: uint64, nSamples: csize_t,
                      outp: ptr uint64): cint {.importc.}
proc cp2_dataset_build(ctx: Cp2Ctx, cfg: ptr Cp2Config, firstSlot, nLocal: uint64, ds: ptr Cp2Dataset): cint {.importc.}
proc cp2_dataset_free(ds: Cp2Dataset) {.importc.}
proc cp2_dataset_build_streamed(ctx: Cp2Ctx, cfg: ptr Cp2Config, firstSlot, nLocal: uint64, entropy: ptr byte, threads: cint,
                                groupSlots: csize_t, ds: ptr Cp2Dataset): cint {.importc.}
proc cp2_dataset_export_streamed(ds: Cp2Dataset, dir: cstring, threads: cint, totalBytes: ptr uint64): cint {.importc.}
proc cp2_proof_input_generate(ds: Cp2Dataset, slotIdx: uint64, entropy: ptr byte, p: ptr Cp2ProofInput): cint {.importc.}
proc cp2_proof_input_free(p: Cp2ProofInput) {.importc.}
proc cp2_proof_input_write_json(p: Cp2ProofInput, path: cstring): cint {.importc.}
proc cp2_proof_input_roots(p: Cp2ProofInput, datasetRoot, slotRoot, entropy: ptr byte): cint {.importc.}
proc cp2_proof_input_nsamples(p: Cp2ProofInput): csize_t {.importc.}
proc cp2_proof_input_cell_indices(p: Cp2ProofInput): ptr UncheckedArray[uint64] {.importc.}
proc cp2_proof_input_cell_data(p: Cp2ProofInput): ptr UncheckedArray[byte] {.importc.}
proc cp2_proof_input_merkle_paths(p: Cp2ProofInput): ptr UncheckedArray[byte] {.importc.}
proc cp2_proof_input_slot_proof(p: Cp2ProofInput): ptr UncheckedArray[byte] {.importc.}
proc cp2_proof_input_leaf_hashes(p: Cp2ProofInput): ptr UncheckedArray[byte] {.importc.}
proc cp2_proof_input_create(cfg: ptr Cp2Config, slotIdx: uint64, datasetRoot, entropy, slotRoot, slotProof: ptr byte,
                            nSamples: csize_t, cellIndices: ptr uint64, cellData, merklePaths, leafHashes: ptr byte,
                            p: ptr Cp2ProofInput): cint {.importc.}
proc cp2_write_circom_main(cfg: ptr Cp2Config, path: cstring): cint {.importc.}
proc cp2_multi_init(devices: ptr cint, nDev: cint, m: ptr Cp2Multi): cint {.importc.}
proc cp2_multi_free(m: Cp2Multi) {.importc.}
proc cp2_multi_count(m: Cp2Multi): cint {.importc.}
proc cp2_multi_ctx(m: Cp2Multi, i: cint): Cp2Ctx {.importc.}
proc cp2_multi_last_error(m: Cp2Multi): cstring {.importc.}
proc cp2_multi_gather_mode(m: Cp2Multi): cstring {.importc.}
proc cp2_multi_set_policy(m: Cp2Multi, gather: cint, minCellsPerDevice: uint64): cint {.importc.}
proc cp2_multi_set_split(m: Cp2Multi, unitsPerSlot: int64): cint {.importc.}
proc cp2_multi_dataset_build(m: Cp2Multi, cfg: ptr Cp2Config, ds: ptr Cp2MultiDataset): cint {.importc.}
proc cp2_multi_dataset_build_cached(m: Cp2Multi, cfg: ptr Cp2Config, cachePath: cstring, ds: ptr Cp2MultiDataset): cint {.importc.}
proc cp2_multi_dataset_build_streamed(m: Cp2Multi, cfg: ptr Cp2Config, entropy: ptr byte, threads: cint, groupSlots: csize_t,
                                      ds: ptr Cp2MultiDataset): cint {.importc.}
proc cp2_check_environment(msg: cstring, msgLen: csize_t): cint {.importc.}
proc cp2_multi_dataset_free(ds: Cp2MultiDataset) {.importc.}
proc cp2_multi_dataset_shards(ds: Cp2MultiDataset): cint {.importc.}
proc cp2_multi_proof_input_generate(ds: Cp2MultiDataset, slotIdx: uint64, entropy: ptr byte, p: ptr Cp2ProofInput): cint {.importc.}
proc cp2_multi_dataset_export_streamed(ds: Cp2MultiDataset, dir: cstring, threads: cint, totalBytes: ptr uint64): cint {.importc.}
{.pop.}

var gMulti: Cp2Multi

proc requireAbi() =
  ## The entry points are bound by name when the library is loaded: only this number tells a library built from another header.
  ## Another MAJOR is refused outright, an older MINOR too (an entry point this binding imports may be missing).
  let v = int(cp2_abi_version())
  let (major, minor) = (v shr 16, v and 0xffff)
  if major != abiVersionMajor or minor < abiVersionMinor:
    raiseAssert("libcodex_p2.so has ABI version " & $major & "." & $minor & ", this binding was written against " &
                $abiVersionMajor & "." & $abiVersionMinor & " (include/codex_p2.h, CP2_ABI_VERSION_*)")

proc multi(): Cp2Multi =
  ## one engine per process (the reference is single threaded, cli.nim:208-237): the seam calls run on the first device's
  ## context, generateProofInput cuts the dataset's slots over all the engine's devices.  The environment variable
  ## CODEX_P2_GPUS ("all", "<count>" or an index list) names the devices -- unset: ONE device, several are opt-in (INTEGRATION.md
  ## section 1); nothing in cli.nim changes.
  if pointer(gMulti) == nil:
    requireAbi()
    let st = cp2_multi_init(nil, 0, addr gMulti)
    if st != 0:
      var why: array[512, char]            # a CODEX_P2_* variable that does not hold what it takes is named, not guessed at
      discard cp2_check_environment(cast[cstring](addr why[0]), csize_t(len(why)))
      raiseAssert("cp2_multi_init: " & $cp2_strerror(st) & " " & $cast[cstring](addr why[0]))
  gMulti

proc ctx(): Cp2Ctx =
  let c = cp2_multi_ctx(multi(), 0)
  if pointer(c) == nil: raiseAssert("cp2_init: " & $cp2_strerror(-2))
  c

proc check(st: cint, what: string) =
  ## nothing aborts across the C ABI; keep the reference's behaviour (assert -> AssertionDefect) on this side
  if st != 0: raiseAssert(what & ": " & $cp2_strerror(st) & " " & $cp2_multi_last_error(multi()) & " " & $cp2_last_error(ctx()))

# Empty inputs are legal at the seam (nim-poseidon2 hashes the padding of an empty sequence, and so does the C ABI when
# the length is 0), but `unsafeAddr a[0]` of an empty openArray raises IndexDefect: hand the engine a valid dummy address.
var gDummy: array[32, byte]
template firstByte[T](a: openArray[T]): ptr byte =
  (if a.len > 0: cast[ptr byte](unsafeAddr a[0]) else: addr gDummy[0])

# ---- the nim-poseidon2 / constantine names the reference imports -------------------------------------
const zero*: F = default(F)

func toF*(x: int): F =
  ## poseidon2/io toF (types/bn254.nim:27, sample/bn254.nim:22)
  var v = uint64(x)
  for i in 0 ..< 8: result[i] = byte((v shr (8 * i)) and 0xff)

proc compress*(x, y: F, key: F = zero): F =
  ## poseidon2/compress (merkle/bn254.nim:18,50,53); key is one of 0,1,2,3
  var xy: array[64, byte]
  for i in 0 ..< 32: (xy[i] = x[i]; xy[32 + i] = y[i])
  check(cp2_compress_batch(ctx(), addr xy[0], uint32(key[0]), addr result[0], 1), "compress")

type Sponge* = object
type Merkle* = object

proc digest*(_: type Sponge, input: openArray[F], rate: static int = 2): F =
  ## Sponge.digest(seq[F], rate = 2)  (sample/bn254.nim:23)
  static: doAssert rate == 2
  check(cp2_sponge2_felts(ctx(), firstByte(input), csize_t(input.len), addr result[0]), "Sponge.digest")

proc digest*(_: type Sponge, input: openArray[byte], rate: static int = 2): F =
  ## Sponge.digest(bytes, rate = 2)  (blocks/bn254.nim:27): 10* byte padding, 31-byte chunks, rate-2 sponge
  static: doAssert rate == 2
  check(cp2_hash_cells(ctx(), firstByte(input), csize_t(input.len), 1, addr result[0]), "Sponge.digest(bytes)")

proc hashCells*(data: openArray[byte], cellSize: int): seq[F] =
  ## data.len / cellSize cells hashed in one launch (blocks/bn254.nim:23-29 applied to every cell of a block)
  doAssert cellSize > 0 and data.len mod cellSize == 0
  result = newSeq[F](data.len div cellSize)
  if result.len > 0:
    check(cp2_hash_cells(ctx(), firstByte(data), csize_t(cellSize), csize_t(result.len), cast[ptr byte](addr result[0])), "hashCells")

proc digest*(_: type Merkle, xs: openArray[F]): F =
  ## Merkle.digest (merkle/bn254.nim:20); an empty input is refused by the engine (Merkle.hs:72 "input is empty") -> raiseAssert
  check(cp2_merkle_root(ctx(), firstByte(xs), csize_t(xs.len), addr result[0]), "Merkle.digest")

iterator elements*(bytes: openArray[byte], _: type F): F =
  ## poseidon2/io elements (json/bn254.nim:11,25)
  let n = int(cp2_felts_per_bytes(csize_t(bytes.len)))
  var buf = newSeq[F](n)            # n >= 1: even empty input yields the chunk that carries the 0x01 marker
  check(cp2_bytes_to_felts(firstByte(bytes), csize_t(bytes.len), cast[ptr byte](addr buf[0])), "elements")
  for f in buf: yield f

func toDecimal*(a: F): string =
  ## constantine io_fields.toDecimal (types/bn254.nim:30): base-10 digits of the canonical integer
  var w: array[8, uint32]
  for i in 0 ..< 8:
    w[i] = uint32(a[4*i]) or (uint32(a[4*i+1]) shl 8) or (uint32(a[4*i+2]) shl 16) or (uint32(a[4*i+3]) shl 24)
  var digits: seq[char]
  var nonzero = true
  while nonzero:
    var rem: uint64 = 0
    nonzero = false
    for i in countdown(7, 0):
      let cur = (rem shl 32) or uint64(w[i])
      w[i] = uint32(cur div 10)
      rem = cur mod 10
      if w[i] != 0: nonzero = true
    digits.add(char(ord('0') + int(rem)))
  for i in countdown(digits.high, 0): result.add(digits[i])

func bit*(a: F, i: int): uint64 = uint64((a[i shr 3] shr (i and 7)) and 1)   ## constantine `bit` (types/bn254.nim:51)
func toBig*(a: F): F = a                                                      ## already canonical

# ---- the whole path as plain Nim values (used by overlay/src/gen_input/bn254.nim and overlay/src/json/bn254.nim) ------
type EngineProofInput* = object
  ## what cp2_proof_input's accessors return, copied into GC-owned Nim values
  dataSetRoot*, slotRoot*, entropy*: F
  slotProof*:   seq[F]            ## maxLog2NSlots elements, zero padded
  cellIndices*: seq[int]
  cellData*:    seq[seq[byte]]    ## nSamples cells
  merklePaths*: seq[seq[F]]       ## nSamples x maxDepth, zero padded
  leafHashes*:  seq[F]            ## hash of each sampled cell

proc felt(p: ptr UncheckedArray[byte], i: int): F =
  for k in 0 ..< 32: result[k] = p[32 * i + k]

proc engineGenerateProofInput*(cfg: var Cp2Config, slotIdx: int, entropy: F): EngineProofInput =
  ## every slot tree built once, the slots cut over all GPUs of the node (one device-to-device gather of the slot roots,
  ## the dataset tree on every device), then sampling + paths + cells for `slotIdx` on the device that holds it
  ## (the whole of gen_input/bn254.nim:35-74)
  ## CODEX_P2_CACHE=<file> (not in the reference, which recomputes every tree on every run): what the datasets keep of their
  ## trees is read from / written to that file (one file per shard), so a later run with new entropy hashes nothing again.
  var ds: Cp2MultiDataset
  let cache = getEnv("CODEX_P2_CACHE")
  if cache.len > 0:
    check(cp2_multi_dataset_build_cached(multi(), addr cfg, cstring(cache), addr ds), "cp2_multi_dataset_build_cached")
  else:
    check(cp2_multi_dataset_build(multi(), addr cfg, addr ds), "cp2_multi_dataset_build")
  defer: cp2_multi_dataset_free(ds)
  var p: Cp2ProofInput
  var e = entropy
  check(cp2_multi_proof_input_generate(ds, uint64(slotIdx), addr e[0], addr p), "cp2_multi_proof_input_generate")
  defer: cp2_proof_input_free(p)
  check(cp2_proof_input_roots(p, addr result.dataSetRoot[0], addr result.slotRoot[0], addr result.entropy[0]), "cp2_proof_input_roots")
  let ns = int(cp2_proof_input_nsamples(p))
  let md = int(cfg.maxDepth)
  let cs = int(cfg.cellSize)
  let sp = cp2_proof_input_slot_proof(p)
  for i in 0 ..< int(cfg.maxLog2NSlots): result.slotProof.add(felt(sp, i))
  let idx = cp2_proof_input_cell_indices(p)
  let cells = cp2_proof_input_cell_data(p)
  let paths = cp2_proof_input_merkle_paths(p)
  let leaves = cp2_proof_input_leaf_hashes(p)
  for i in 0 ..< ns:
    result.cellIndices.add(int(idx[i]))
    var cell = newSeq[byte](cs)
    for k in 0 ..< cs: cell[k] = cells[i * cs + k]
    result.cellData.add(cell)
    var path = newSeq[F](md)
    for d in 0 ..< md: path[d] = felt(paths, i * md + d)
    result.merklePaths.add(path)
    result.leafHashes.add(felt(leaves, i))

proc engineWriteProofInputJson*(cfg: var Cp2Config, slotIdx: int, v: EngineProofInput, fname: string) =
  ## json/bn254.nim:57-74 through the engine's byte-exact writer, from plain values
  let ns = v.cellData.len
  let md = int(cfg.maxDepth)
  let cs = int(cfg.cellSize)
  var proof = newSeq[byte](max(1, v.slotProof.len * 32))
  for i in 0 ..< v.slotProof.len:
    for k in 0 ..< 32: proof[32 * i + k] = v.slotProof[i][k]
  var idx = newSeq[uint64](max(1, ns))
  var cells = newSeq[byte](max(1, ns * cs))
  var paths = newSeq[byte](max(1, ns * md * 32))
  for i in 0 ..< ns:
    if i < v.cellIndices.len: idx[i] = uint64(v.cellIndices[i])
    doAssert v.cellData[i].len == cs and v.merklePaths[i].len == md
    for k in 0 ..< cs: cells[i * cs + k] = v.cellData[i][k]
    for d in 0 ..< md:
      for k in 0 ..< 32: paths[(i * md + d) * 32 + k] = v.merklePaths[i][d][k]
  var dr = v.dataSetRoot
  var en = v.entropy
  var sr = v.slotRoot
  var p: Cp2ProofInput
  check(cp2_proof_input_create(addr cfg, uint64(slotIdx), addr dr[0], addr en[0], addr sr[0], addr proof[0], csize_t(ns),
                               addr idx[0], addr cells[0], addr paths[0], nil, addr p), "cp2_proof_input_create")
  defer: cp2_proof_input_free(p)
  check(cp2_proof_input_write_json(p, cstring(fname)), "cp2_proof_input_write_json")

proc engineMerkleLayers*(xs: openArray[F]): seq[seq[F]] =
  ## all layers, bottom first (merkle/bn254.nim:24-63), through cp2_merkle_tree
  doAssert xs.len > 0
  let total = int(cp2_merkle_total(csize_t(xs.len)))
  var flat = newSeq[F](total)
  var sizes = newSeq[csize_t](80)
  var nl: csize_t
  check(cp2_merkle_tree(ctx(), firstByte(xs), csize_t(xs.len), cast[ptr byte](addr flat[0]),
                        addr sizes[0], addr nl), "cp2_merkle_tree")
  var off = 0
  for k in 0 ..< int(nl):
    result.add(flat[off ..< off + int(sizes[k])])
    off += int(sizes[k])

proc engineCellIndices*(entropy, slotRoot: F, numberOfCells, nSamples: int): seq[int] =
  ## sample/bn254.nim:16-27 for counters 1..nSamples in one launch
  var raw = newSeq[uint64](max(1, nSamples))
  var e = entropy
  var r = slotRoot
  check(cp2_cell_indices(ctx(), addr e[0], addr r[0], uint64(numberOfCells), csize_t(nSamples), addr raw[0]), "cp2_cell_indices")
  for i in 0 ..< nSamples: result.add(int(raw[i]))

proc writeCircomMainComponentP2*(cfg: var Cp2Config, fname: string) =
  check(cp2_write_circom_main(addr cfg, cstring(fname)), "cp2_write_circom_main")

proc engineExportAllProofInputs*(cfg: var Cp2Config, entropy: F, dir: string, threads: int = 8): uint64 =
  ## every slot's input.json ("<dir>/input_<slot>.json") in ONE overlapped pass (no reference counterpart: the reference makes
  ## one proof input per run): gen_input/bn254.nim:35-79 + json/bn254.nim:57-78 for all slots, trees built once
  var ds: Cp2MultiDataset
  var e = entropy
  check(cp2_multi_dataset_build_streamed(multi(), addr cfg, addr e[0], cint(threads), 0, addr ds), "cp2_multi_dataset_build_streamed")
  defer: cp2_multi_dataset_free(ds)
  check(cp2_multi_dataset_export_streamed(ds, cstring(dir), cint(threads), addr result), "cp2_multi_dataset_export_streamed")

proc engineDevices*(): int =
  ## how many devices the engine holds
  int(cp2_multi_count(multi()))

proc engineGatherMode*(): string =
  ## what the last build's exchange of slot roots went through: "rccl (...)", "host (<why>)", "copy (...)", "none (one shard ...)"
  $cp2_multi_gather_mode(multi())

proc engineSetSplit*(unitsPerSlot: int) =
  ## units every slot is cut into when a dataset of few, large slots is spread over several GPUs: 0 choose, 1 whole slots only
  check(cp2_multi_set_split(multi(), int64(unitsPerSlot)), "cp2_multi_set_split")

proc engineSetPolicy*(gather: int, minCellsPerDevice: uint64) =
  ## 0 auto / 1 RCCL / 2 host gather / 3 device-to-device copies; cells of hashing a device must have to get a shard (0: one hash-kernel residency)
  check(cp2_multi_set_policy(multi(), cint(gather), minCellsPerDevice), "cp2_multi_set_policy")

iterator contexts(): Cp2Ctx =
  ## the context of every device the engine holds (the per-context knobs below are set on all of them)
  for i in 0 ..< int(cp2_multi_count(multi())):
    let c = cp2_multi_ctx(multi(), cint(i))
    if pointer(c) == nil: raiseAssert("cp2_multi_ctx: device " & $i & " is unusable")
    yield c

proc engineTrim*() =
  ## give the engine's cached device / pinned scratch back to the system (a long-lived process between runs)
  for c in contexts(): check(cp2_trim(c), "cp2_trim")

proc engineSetIngestDirect*(on: bool) =
  ## SlotFile source: O_DIRECT reads of slot files that are not in the page cache
  for c in contexts(): check(cp2_set_ingest_direct(c, cint(ord(on))), "cp2_set_ingest_direct")

proc engineSetIngestMapped*(on: bool) =
  ## SlotFile source, opt-in: chunks of slot files that sit in the page cache are uploaded from registered windows of the file's
  ## mapping instead of being copied into the pinned ring by host threads (same throughput on one device, fewer busy cores)
  for c in contexts(): check(cp2_set_ingest_mapped(c, cint(ord(on))), "cp2_set_ingest_mapped")

proc engineSetKeepTrees*(mode: int) =
  ## what a dataset keeps of its slot trees in device memory: 1 every node; 2 compact (block roots and up, the bottom of a path
  ## recomputed from the touched blocks); 0 roots only (the proved slot's tree is rebuilt on demand); -1 the most that fits
  for c in contexts(): check(cp2_set_keep_trees(c, cint(mode)), "cp2_set_keep_trees")

proc engineSetBodyBudget*(maxResidentBytes: int, spillDir: string = "") =
  for c in contexts():
    check(cp2_set_body_budget(c, csize_t(maxResidentBytes), (if spillDir.len > 0: cstring(spillDir) else: nil)), "cp2_set_body_budget")
