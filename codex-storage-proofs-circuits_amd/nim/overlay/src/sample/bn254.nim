## OVERLAY of reference/nim/proof_input/src/sample/bn254.nim (cellIndex, cellIndices).  Uncompiled; mechanical.
import ../types
import ../types/bn254
import ../misc
import ../codex_p2

proc cellIndices*(hashcfg: HashConfig, entropy: Entropy, slotRoot: Root, numberOfCells: int, nSamples: int): seq[int] =
  ## sample/bn254.nim:26-27 (counters 1..nSamples), one GPU launch for all of them
  assert hashcfg.field == BN254
  let log2 = ceilingLog2(numberOfCells)
  assert (1 shl log2) == numberOfCells, "for this version, `numberOfCells` is assumed to be a power of two"
  engineCellIndices(entropy, slotRoot, numberOfCells, nSamples)

proc cellIndex*(hashcfg: HashConfig, entropy: Entropy, slotRoot: Root, numberOfCells: int, counter: int): int =
  ## sample/bn254.nim:16-24
  cellIndices(hashcfg, entropy, slotRoot, numberOfCells, counter)[counter - 1]
