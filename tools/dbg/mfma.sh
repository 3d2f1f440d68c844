hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_peak tools/mb/mfma_peak.hip
timeout -k 5 60 /tmp/mfma_peak 8 100
timeout -k 5 60 /tmp/mfma_peak 4 100
timeout -k 5 60 /tmp/mfma_peak 8 1000 | tail -2
