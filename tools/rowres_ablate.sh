for ab in 0 128 64; do echo "ABLATE=$ab"; PDN_ROWRES_ABLATE=$ab NO_EPI=1 timeout 300 python tools/rowres_probe.py 65536 2>&1 | grep -E "N=  288|N= 1536|N=32064"; done
