for name in base wl4 cap32 pairs128; do
  lib=""; [ "$name" != base ] && lib=$PWD/pheniqs_amd/variants/lib$name.so
  echo "== $name"; PHQ_LIBRARY=$lib bash profiles/tools/shapes.sh "C3:50000000:estimated"
done
for name in base trip2; do
  lib=""; [ "$name" != base ] && lib=$PWD/pheniqs_amd/variants/lib$name.so
  echo "== $name"; PHQ_LIBRARY=$lib bash profiles/tools/shapes.sh "C2 C3:50000000 S384"
done
