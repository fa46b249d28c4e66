for rep in 1 2 3; do
for v in "512 all" "256 all" "256 chunk" "320 chunk"; do
set -- $v
echo "== chunk $1 read $2"
MELF_JPEG_CHUNK=$1 MELF_JPEG_READ=$2 python3 tools/files_api_rate.py 1024 | tail -1
done; done
