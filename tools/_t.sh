python tools/asm_pts_check.py check 2>&1 | tail -1
for sh in "21 1000 c2xc2" "9 2000 c3xc2" "12 1500 c3xc2"; do
for o in "" "asm.pts_debug=3" "asm.pts_debug=2"; do
echo "shape $sh opts: $o"; python tools/asm_perm_one.py $sh $o 2>&1 | grep "assemble\|rror" | tail -1
done; done
