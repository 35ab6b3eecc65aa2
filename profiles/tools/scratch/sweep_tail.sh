for g in 1 2 4 8 16; do echo "EXACT_GROUP $g"; PHQ_EXACT_GROUP=$g python profiles/tools/scratch/small_batch.py C2 2>&1 | grep -v amdgpu; done
for s in 8 12 16 24; do echo "SHARE_NEAR $s"; PHQ_GRID_SHARE_NEAR=$s bash profiles/tools/shapes.sh "U384 S384 L12"; done
for s in 4 8 16; do echo "SHARE_CONSUMER $s"; PHQ_GRID_SHARE_CONSUMER=$s bash profiles/tools/shapes.sh "U384 S384 H94"; done
