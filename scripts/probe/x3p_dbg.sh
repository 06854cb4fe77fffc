# K-loop probes of the three-plane ring tiles: full kernel, no MFMAs, no refills, neither (A/B builds in build_probe/)
echo "full"; X3P_TILES=18,19,24,25 timeout 300 python scripts/x3p_check.py bench 2>/dev/null | head -1
for d in 1 2 3; do echo "probe=$d"; IPRGAN_LIB=$PWD/build_probe/libprobe$d.so X3P_TILES=18,19,24,25 timeout 300 python scripts/x3p_check.py bench 2>/dev/null | head -1; done
