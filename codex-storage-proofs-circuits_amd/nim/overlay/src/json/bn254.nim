## OVERLAY of reference/nim/proof_input/src/json/bn254.nim: exportProofInputBN254 with the reference's own signature,
## written by the engine's byte-exact writer from the SlotProofInput VALUE (any value, not only engine-made ones).
## Uncompiled; mechanical by design.
import ../types
import ../types/bn254
import ../codex_p2

proc exportProofInput*(fname: string, prfInput: SlotProofInput[Hash]) =
  var cfg: Cp2Config
  cfg.maxLog2NSlots = int32(prfInput.slotProof.merklePath.len)
  cfg.maxDepth = if prfInput.proofInputs.len > 0: int32(prfInput.proofInputs[0].merkleProof.merklePath.len) else: 0
  cfg.cellSize = if prfInput.proofInputs.len > 0: uint64(prfInput.proofInputs[0].cellData.len) else: 0
  cfg.nSlots = uint64(prfInput.nSlots)
  cfg.nCells = uint64(prfInput.nCells)
  var v: EngineProofInput
  v.dataSetRoot = prfInput.dataSetRoot
  v.entropy = prfInput.entropy
  v.slotRoot = prfInput.slotRoot
  v.slotProof = prfInput.slotProof.merklePath
  for p in prfInput.proofInputs:
    v.cellIndices.add(p.merkleProof.leafIndex)
    v.cellData.add(p.cellData)
    v.merklePaths.add(p.merkleProof.merklePath)
  engineWriteProofInputJson(cfg, prfInput.slotIndex, v, fname)

proc exportProofInputBN254*(hashcfg: HashConfig, fname: string, prfInput: SlotProofInput[Hash]) =
  assert hashcfg.field == BN254
  exportProofInput(fname, prfInput)
