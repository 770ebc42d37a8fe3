for f in 0 256 512 1024 1792; do echo "== relu2=$f (dbg $((f>>8)))"; timeout -k 10 120 python scripts/check_wzy.py --no-check --relu2 $f --shapes "4,192,64,128" 2>&1 | tail -1 | cut -c1-160; done
