#include "badger_hip.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv)
{
    for (int rep = 0; rep < 3; ++rep) {
        bdg_ingest* g = nullptr;
        int rc = bdg_ingest_open_mt(argv[1], 256, 3, 0, (uint32_t)atoi(argv[2]), &g);
        if (rc) { printf("open rc %d\n", rc); return 1; }
        unsigned long long n = 0, bytes = 0; int k = 0;
        for (;;) {
            bdg_ingest_chunk ch;
            rc = bdg_ingest_next(g, &ch);
            if (rc) { printf("next rc %d: %s\n", rc, bdg_ingest_error(g)); break; }
            if (ch.n == 0) break;
            n += ch.n; bytes += ch.total_bytes;
            bdg_ingest_release(g, ch.id);
            if (rep == 2 && ++k == 2) break;          // close in mid-file: threads must wind down
        }
        bdg_ingest_close(g);
        printf("reads %llu bytes %llu\n", n, bytes);
    }
    return 0;
}
