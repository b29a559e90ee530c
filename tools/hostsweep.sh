for p in 4 8 12 16 24 32; do echo "pieces=$p"; C25519_AMD_BATCH_PIECES=$p python tools/hostapi_rate.py 2>&1 | grep -E "^x25519|^sign|^verify" | cut -c1-100; done
