#!/bin/bash
# VGPR / scratch / LDS of every kernel matching $1 (default: fefp_kernel), from the compiler's own remarks (no GPU needed).
cd "$(dirname "$0")/../dolfinx_materials_amd/csrc" || exit 1
make -s asm 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|VGPRs Spill|LDS Size" | grep -A4 "${1:-fefp_kernel}" | grep -v "^--" | paste - - - - - \
  | sed -E 's/.*Function Name: _ZN3dxm[0-9]+([a-z_0-9]+)I([A-Za-z0-9]+)EEv.*VGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*Spill: ([0-9]+).*LDS Size \[bytes\/block\]: ([0-9]+).*/\1<\2> vgpr=\3 scratch=\4 spill=\5 lds=\6/'
