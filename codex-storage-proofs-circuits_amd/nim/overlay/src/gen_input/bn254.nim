## OVERLAY of reference/nim/proof_input/src/gen_input/bn254.nim: generateProofInputBN254 with the reference's own
## signature and result type, computed by the MI355X engine (every slot tree built ONCE; the reference builds all of
## them and then the proving slot's again per sample, gen_input/bn254.nim:42,57) on EVERY GPU of the node: the slots are cut
## into contiguous ranges, one per device, the slot roots gathered device to device (RCCL over xGMI) and the dataset tree
## built on every device (cp2_multi_*, include/codex_p2.h section e).  cli.nim needs no change.
## Uncompiled (no Nim toolchain in the build image); mechanical by design.
import ../types
import ../types/bn254
import ../codex_p2

proc toEngineConfig(globCfg: GlobalConfig, dsetCfg: DataSetConfig): Cp2Config =
  result.maxDepth = int32(globCfg.maxDepth)
  result.maxLog2NSlots = int32(globCfg.maxLog2NSlots)
  result.cellSize = uint64(globCfg.cellSize)
  result.blockSize = uint64(globCfg.blockSize)
  result.nSlots = uint64(dsetCfg.nSlots)
  result.nCells = uint64(dsetCfg.nCells)
  result.nSamples = uint64(dsetCfg.nSamples)
  case dsetCfg.dataSrc.kind
  of FakeData:
    result.seed = dsetCfg.dataSrc.seed
    result.fileBase = nil
  of SlotFile:
    result.fileBase = cstring(dsetCfg.dataSrc.filename)      # slot k = "<base><k>.dat", dataset.nim:34

proc generateProofInput*(hashCfg: HashConfig, globCfg: GlobalConfig, dsetCfg: DataSetConfig, slotIdx: SlotIdx,
                         entropy: Entropy): SlotProofInput[Hash] =
  assert hashCfg.field == BN254
  assert dsetCfg.nCells mod cellsPerBlock(globCfg) == 0
  var cfg = toEngineConfig(globCfg, dsetCfg)
  let v = engineGenerateProofInput(cfg, slotIdx, entropy)
  var inputs: seq[CellProofInput[Hash]]
  for i in 0 ..< v.cellData.len:
    let prf = MerkleProof[Hash](leafIndex: v.cellIndices[i], leafValue: v.leafHashes[i], merklePath: v.merklePaths[i],
                                numberOfLeaves: dsetCfg.nCells)     # merged + padded, merkle.nim:86-100, types.nim:27-37
    inputs.add(CellProofInput[Hash](cellData: v.cellData[i], merkleProof: prf))
  return SlotProofInput[Hash](dataSetRoot: v.dataSetRoot
                             , entropy:     v.entropy
                             , nCells:      dsetCfg.nCells
                             , nSlots:      dsetCfg.nSlots
                             , slotIndex:   slotIdx
                             , slotRoot:    v.slotRoot
                             , slotProof:   MerkleProof[Hash](leafIndex: slotIdx, leafValue: v.slotRoot,
                                                              merklePath: v.slotProof, numberOfLeaves: dsetCfg.nSlots)
                             , proofInputs: inputs
                             )

proc generateProofInputBN254*(hashCfg: HashConfig, globCfg: GlobalConfig, dsetCfg: DataSetConfig, slotIdx: SlotIdx,
                              entropy: Entropy): SlotProofInput[Hash] =
  generateProofInput(hashCfg, globCfg, dsetCfg, slotIdx, entropy)
