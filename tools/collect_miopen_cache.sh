#!/bin/bash
# Run on the GPU box after a bench: copy the in-tree MIOpen cache into gpurun_out/ so it comes back.
mkdir -p gpurun_out/miopen_tree && cp -r .miopen/* gpurun_out/miopen_tree/ 2>/dev/null; du -sh gpurun_out/miopen_tree
