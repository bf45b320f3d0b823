/* mn_nodemap.h — shared by the graph SQL entry points: identifier validation and the string → first-seen-index map
 * (a hash map where the reference scans linearly: src/node2vec.c:72-77, src/graph_tvf.c:1231-1235,1591-1595). */
#ifndef MN_NODEMAP_H
#define MN_NODEMAP_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define MN_UNUSED __attribute__((unused))

/* MUNINN_DEVICE=<HIP ordinal> pins this process's extension to one GPU of the node (default 0): with one process per GPU a
 * host serves independent connections from all of them; the jointly built / sharded multi-GPU modes are the C-ABI's
 * mn_comm entry points (include/muninn_hip.h "multi-GPU"). */
MN_UNUSED static int mn_env_device(void) {
    const char *e = getenv("MUNINN_DEVICE");
    if (!e)
        return 0;
    char *end = NULL;
    long d = strtol(e, &end, 10);
    if (end == e || *end != '\0' || d < 0 || d > 1023)
        return -1; /* malformed ("abc", "", "1x", "-1"): no device has this ordinal, so creation fails and says so */
    return (int)d;
}
/* for error messages about a device: " (MUNINN_DEVICE='...')" when the variable is set, else "" (static buffer, one thread's use) */
MN_UNUSED static const char *mn_env_device_hint(void) {
    static _Thread_local char buf[96];
    const char *e = getenv("MUNINN_DEVICE");
    if (!e)
        return "";
    snprintf(buf, sizeof(buf), " (MUNINN_DEVICE='%.60s')", e);
    return buf;
}

/* ───────────────────────── shared: identifiers, string→index map ───────────────────────── */

MN_UNUSED static int ident_ok(const char *s) { /* id_validate, src/id_validate.c:17-28 */
    if (!s || !*s)
        return 0;
    for (; *s; s++)
        if (!((*s >= 'a' && *s <= 'z') || (*s >= 'A' && *s <= 'Z') || (*s >= '0' && *s <= '9') || *s == '_'))
            return 0;
    return 1;
}

typedef struct {
    char **ids;
    int n, cap;
    int *slots; /* open addressing: slot → node index, -1 empty */
    int nslots;
} NodeMap;

MN_UNUSED static unsigned long djb2(const char *s) { /* src/graph_common.h:33-38 */
    unsigned long h = 5381;
    for (; *s; s++)
        h = ((h << 5) + h) + (unsigned char)*s;
    return h;
}

MN_UNUSED static void nm_init(NodeMap *m) {
    m->n = 0;
    m->cap = 256;
    m->ids = (char **)calloc((size_t)m->cap, sizeof(char *));
    m->nslots = 1024;
    m->slots = (int *)malloc((size_t)m->nslots * sizeof(int));
    for (int i = 0; m->slots && i < m->nslots; i++)
        m->slots[i] = -1;
}

MN_UNUSED static void nm_free(NodeMap *m) {
    for (int i = 0; m->ids && i < m->n; i++)
        free(m->ids[i]);
    free(m->ids);
    free(m->slots);
}

/* first-seen index, as graph_node_index / graph_data_find_or_add; -1 = out of memory (the map stays consistent) */
MN_UNUSED static int nm_get(NodeMap *m, const char *id) {
    if (!m->ids || !m->slots)
        return -1;
    unsigned long h = djb2(id);
    for (int i = 0;; i++) {
        int s = (int)((h + (unsigned long)i) & (unsigned long)(m->nslots - 1));
        if (m->slots[s] < 0) {
            if (m->n >= m->cap) {
                char **ni = (char **)realloc(m->ids, (size_t)m->cap * 2 * sizeof(char *));
                if (!ni)
                    return -1;
                m->ids = ni;
                m->cap *= 2;
            }
            size_t len = strlen(id) + 1;
            char *copy = (char *)malloc(len);
            if (!copy)
                return -1;
            memcpy(copy, id, len);
            if ((m->n + 1) * 10 > m->nslots * 7) { /* rehash first, so that a failure leaves the map as it was */
                int ns = m->nslots * 2;
                int *nsl = (int *)malloc((size_t)ns * sizeof(int));
                if (!nsl) {
                    free(copy);
                    return -1;
                }
                for (int k = 0; k < ns; k++)
                    nsl[k] = -1;
                for (int k = 0; k < m->n; k++) {
                    unsigned long hk = djb2(m->ids[k]);
                    for (int j = 0;; j++) {
                        int t = (int)((hk + (unsigned long)j) & (unsigned long)(ns - 1));
                        if (nsl[t] < 0) {
                            nsl[t] = k;
                            break;
                        }
                    }
                }
                free(m->slots);
                m->slots = nsl;
                m->nslots = ns;
                for (int j = 0;; j++) { /* the new id's slot in the grown table */
                    s = (int)((h + (unsigned long)j) & (unsigned long)(ns - 1));
                    if (m->slots[s] < 0)
                        break;
                }
            }
            m->ids[m->n] = copy;
            m->slots[s] = m->n++;
            return m->n - 1;
        }
        if (!strcmp(m->ids[m->slots[s]], id))
            return m->slots[s];
    }
}

#endif
