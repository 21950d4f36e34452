# the HBM-bound 1x1 layers of the ResidualBlocks at 18 x 256 x 384: tile variants, schedules, stage paths
C="python tools/one_layer.py --kind conv --k 1 --s 1 --n 18 --hw 256 384 --reps 10"
echo "== 96->192 + skip"; $C --cin 96 --cout 192 --epi --ab 1,2,3,6,8,9
echo "== 96->192 + skip, static"; $C --cin 96 --cout 192 --epi --static --ab 2,3,6,8
echo "== 96->192 + skip, dma"; $C --cin 96 --cout 192 --epi --dma 1 --ab 2,3,8,9
echo "== 96->192 no skip"; $C --cin 96 --cout 192 --ab 3,6
echo "== 192->96"; $C --cin 192 --cout 96 --ab 1,3,8
echo "== 192->96 static"; $C --cin 192 --cout 96 --static --ab 1,3,8
echo "== 192->96 dma"; $C --cin 192 --cout 96 --dma 1 --ab 1,3,8
