## OVERLAY of reference/nim/proof_input/src/types/bn254.nim: same module path, same exported names, field
## elements are codex_p2.F (32-byte canonical little-endian) instead of constantine's Fr.  Uncompiled (no Nim
## toolchain in the build image); mechanical by design.
import std/strutils
import std/streams

import ../codex_p2
export codex_p2.F, codex_p2.toF, codex_p2.zero

type BN254_T* = F
type Entropy* = F
type Hash*    = F
type Root*    = Hash

func intToBN254*(x: int): F = toF(x)

func toDecimalF*(a: F): string =
  ## types/bn254.nim:29-37: canonical decimal, leading zeros stripped, "0" for zero
  var s = toDecimal(a)
  s = s.strip(leading = true, trailing = false, chars = {'0'})
  if s.len == 0: s = "0"
  return s

func toQuotedDecimalF*(x: F): string = "\"" & toDecimalF(x) & "\""

proc writeLnF*(h: Stream, prefix: string, x: F) = h.writeLine(prefix & toQuotedDecimalF(x))
proc writeF*(h: Stream, prefix: string, x: F) = h.write(prefix & toQuotedDecimalF(x))

func extractLowBits*(fld: F, k: int): uint64 =
  ## types/bn254.nim:47-59: the k low bits of the canonical representative
  assert k > 0 and k <= 64
  for i in 0 ..< k:
    if bit(fld, i) != 0: result = result or (1'u64 shl i)
