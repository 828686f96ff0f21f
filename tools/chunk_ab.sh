for c in 32 64 96 128 192 256; do echo "chunk $c"; ANX_SCAN_CHUNK_FUSED=$c bash tools/quick_ab.sh tree 2>&1 | tail -1; done
