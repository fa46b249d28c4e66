# files API rate against the size of the I/O pool (MELF_IO_THREADS: file reads + header parse + Huffman decode data)
for rep in 1 2; do
for io in 6 8 12 16; do
printf "io %2d: " $io
MELF_IO_THREADS=$io python3 tools/files_api_rate.py ${1:-1024} | tail -1
done; done
