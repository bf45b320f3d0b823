/*
 * mn_sqlite_abi.h — the slice of SQLite's loadable-extension ABI this project uses, written out
 * by hand so the extension builds where no SQLite development headers are installed.
 *
 * SQLite's extension ABI is append-only and public (https://sqlite.org/loadext.html,
 * https://sqlite.org/vtab.html): an extension receives a pointer to `sqlite3_api_routines`, a
 * struct of function pointers whose member ORDER never changes.  Only the slot NUMBERS of the
 * routines used here are recorded (MN_SLOT_*); tests/test_sqlite_ext.py (test_abi_header_matches_real_sqlite_header) re-derives them — and the
 * vtab struct layouts below — from a real sqlite3ext.h when one is available and fails on any
 * drift.  All routines used exist since SQLite 3.8.2 (estimatedRows) or earlier.
 */
#ifndef MN_SQLITE_ABI_H
#define MN_SQLITE_ABI_H

#include <stdarg.h>
#include <stdint.h>

typedef struct sqlite3 sqlite3;
typedef struct sqlite3_stmt sqlite3_stmt;
typedef struct sqlite3_value sqlite3_value;
typedef struct sqlite3_context sqlite3_context;
typedef long long sqlite3_int64;
typedef unsigned long long sqlite3_uint64;

typedef struct sqlite3_api_routines {
    void *slot[272]; /* 2168 bytes in 3.50; slots above 162 are only touched after a version check */
} sqlite3_api_routines;

/* result codes / fundamental types / flags (sqlite.org/rescode.html, c3ref/c_blob.html) */
#define SQLITE_OK 0
#define SQLITE_ERROR 1
#define SQLITE_NOMEM 7
#define SQLITE_CONSTRAINT 19
#define SQLITE_ROW 100
#define SQLITE_DONE 101
#define SQLITE_INTEGER 1
#define SQLITE_FLOAT 2
#define SQLITE_TEXT 3
#define SQLITE_BLOB 4
#define SQLITE_NULL 5
#define SQLITE_UTF8 1
#define SQLITE_DETERMINISTIC 0x000000800
#define SQLITE_INDEX_CONSTRAINT_EQ 2
#define SQLITE_INDEX_CONSTRAINT_MATCH 64
typedef void (*sqlite3_destructor_type)(void *);
#define SQLITE_STATIC ((sqlite3_destructor_type)0)
#define SQLITE_TRANSIENT ((sqlite3_destructor_type)-1)

/* slot numbers inside sqlite3_api_routines */
#define MN_SLOT_bind_blob 2
#define MN_SLOT_bind_double 3
#define MN_SLOT_bind_int 4
#define MN_SLOT_bind_int64 5
#define MN_SLOT_bind_null 6
#define MN_SLOT_bind_text 10
#define MN_SLOT_bind_value 12
#define MN_SLOT_column_blob 19
#define MN_SLOT_column_bytes 20
#define MN_SLOT_column_count 22
#define MN_SLOT_column_double 27
#define MN_SLOT_column_int 28
#define MN_SLOT_column_int64 29
#define MN_SLOT_column_text 36
#define MN_SLOT_column_type 38
#define MN_SLOT_create_function 45
#define MN_SLOT_create_module 47
#define MN_SLOT_declare_vtab 50
#define MN_SLOT_errmsg 53
#define MN_SLOT_exec 55
#define MN_SLOT_finalize 57
#define MN_SLOT_free 58
#define MN_SLOT_last_insert_rowid 65
#define MN_SLOT_libversion_number 67
#define MN_SLOT_malloc 68
#define MN_SLOT_mprintf 69
#define MN_SLOT_reset 77
#define MN_SLOT_result_blob 78
#define MN_SLOT_result_double 79
#define MN_SLOT_result_error 80
#define MN_SLOT_result_int 82
#define MN_SLOT_result_int64 83
#define MN_SLOT_result_null 84
#define MN_SLOT_result_text 85
#define MN_SLOT_step 94
#define MN_SLOT_user_data 101
#define MN_SLOT_value_blob 102
#define MN_SLOT_value_bytes 103
#define MN_SLOT_value_double 105
#define MN_SLOT_value_int 106
#define MN_SLOT_value_int64 107
#define MN_SLOT_value_text 109
#define MN_SLOT_value_type 113
#define MN_SLOT_prepare_v2 116
#define MN_SLOT_create_module_v2 119
#define MN_SLOT_context_db_handle 149
#define MN_SLOT_create_function_v2 162
#define MN_SLOT_set_last_insert_rowid 216 /* SQLite >= 3.18.0; guarded by libversion_number at the call site */

extern const sqlite3_api_routines *mn_sqlite_api; /* set once by sqlite3_muninn_init */
#define MN_API(slotno, fntype) ((fntype)(mn_sqlite_api->slot[slotno]))

/* virtual-table structs (sqlite.org/vtab.html) */
typedef struct sqlite3_vtab sqlite3_vtab;
typedef struct sqlite3_vtab_cursor sqlite3_vtab_cursor;
typedef struct sqlite3_index_info sqlite3_index_info;
typedef struct sqlite3_module sqlite3_module;

struct sqlite3_module {
    int iVersion;
    int (*xCreate)(sqlite3 *, void *pAux, int argc, const char *const *argv, sqlite3_vtab **ppVTab, char **);
    int (*xConnect)(sqlite3 *, void *pAux, int argc, const char *const *argv, sqlite3_vtab **ppVTab, char **);
    int (*xBestIndex)(sqlite3_vtab *pVTab, sqlite3_index_info *);
    int (*xDisconnect)(sqlite3_vtab *pVTab);
    int (*xDestroy)(sqlite3_vtab *pVTab);
    int (*xOpen)(sqlite3_vtab *pVTab, sqlite3_vtab_cursor **ppCursor);
    int (*xClose)(sqlite3_vtab_cursor *);
    int (*xFilter)(sqlite3_vtab_cursor *, int idxNum, const char *idxStr, int argc, sqlite3_value **argv);
    int (*xNext)(sqlite3_vtab_cursor *);
    int (*xEof)(sqlite3_vtab_cursor *);
    int (*xColumn)(sqlite3_vtab_cursor *, sqlite3_context *, int);
    int (*xRowid)(sqlite3_vtab_cursor *, sqlite3_int64 *pRowid);
    int (*xUpdate)(sqlite3_vtab *, int, sqlite3_value **, sqlite3_int64 *);
    int (*xBegin)(sqlite3_vtab *pVTab);
    int (*xSync)(sqlite3_vtab *pVTab);
    int (*xCommit)(sqlite3_vtab *pVTab);
    int (*xRollback)(sqlite3_vtab *pVTab);
    int (*xFindFunction)(sqlite3_vtab *pVtab, int nArg, const char *zName,
                         void (**pxFunc)(sqlite3_context *, int, sqlite3_value **), void **ppArg);
    int (*xRename)(sqlite3_vtab *pVtab, const char *zNew);
    int (*xSavepoint)(sqlite3_vtab *pVTab, int);
    int (*xRelease)(sqlite3_vtab *pVTab, int);
    int (*xRollbackTo)(sqlite3_vtab *pVTab, int);
    int (*xShadowName)(const char *);
    int (*xIntegrity)(sqlite3_vtab *pVTab, const char *zSchema, const char *zTabName, int mFlags, char **pzErr);
};

struct sqlite3_vtab {
    const sqlite3_module *pModule;
    int nRef;
    char *zErrMsg;
};

struct sqlite3_vtab_cursor {
    sqlite3_vtab *pVtab;
};

struct sqlite3_index_info {
    int nConstraint;
    struct sqlite3_index_constraint {
        int iColumn;
        unsigned char op;
        unsigned char usable;
        int iTermOffset;
    } *aConstraint;
    int nOrderBy;
    struct sqlite3_index_orderby {
        int iColumn;
        unsigned char desc;
    } *aOrderBy;
    struct sqlite3_index_constraint_usage {
        int argvIndex;
        unsigned char omit;
    } *aConstraintUsage;
    int idxNum;
    char *idxStr;
    int needToFreeIdxStr;
    int orderByConsumed;
    double estimatedCost;
    sqlite3_int64 estimatedRows;
    int idxFlags;
    sqlite3_uint64 colUsed;
};

/* typed accessors, named as the C API names them */
#define sqlite3_last_insert_rowid MN_API(MN_SLOT_last_insert_rowid, sqlite3_int64 (*)(sqlite3 *))
#define sqlite3_set_last_insert_rowid MN_API(MN_SLOT_set_last_insert_rowid, void (*)(sqlite3 *, sqlite3_int64))
#define sqlite3_libversion_number MN_API(MN_SLOT_libversion_number, int (*)(void))
#define sqlite3_malloc MN_API(MN_SLOT_malloc, void *(*)(int))
#define sqlite3_free MN_API(MN_SLOT_free, void (*)(void *))
#define sqlite3_mprintf MN_API(MN_SLOT_mprintf, char *(*)(const char *, ...))
#define sqlite3_declare_vtab MN_API(MN_SLOT_declare_vtab, int (*)(sqlite3 *, const char *))
#define sqlite3_create_module MN_API(MN_SLOT_create_module, int (*)(sqlite3 *, const char *, const sqlite3_module *, void *))
#define sqlite3_create_module_v2 \
    MN_API(MN_SLOT_create_module_v2, int (*)(sqlite3 *, const char *, const sqlite3_module *, void *, void (*)(void *)))
#define sqlite3_create_function                                                                                         \
    MN_API(MN_SLOT_create_function,                                                                                     \
           int (*)(sqlite3 *, const char *, int, int, void *, void (*)(sqlite3_context *, int, sqlite3_value **),      \
                   void (*)(sqlite3_context *, int, sqlite3_value **), void (*)(sqlite3_context *)))
#define sqlite3_exec \
    MN_API(MN_SLOT_exec, int (*)(sqlite3 *, const char *, int (*)(void *, int, char **, char **), void *, char **))
#define sqlite3_prepare_v2 MN_API(MN_SLOT_prepare_v2, int (*)(sqlite3 *, const char *, int, sqlite3_stmt **, const char **))
#define sqlite3_step MN_API(MN_SLOT_step, int (*)(sqlite3_stmt *))
#define sqlite3_reset MN_API(MN_SLOT_reset, int (*)(sqlite3_stmt *))
#define sqlite3_finalize MN_API(MN_SLOT_finalize, int (*)(sqlite3_stmt *))
#define sqlite3_errmsg MN_API(MN_SLOT_errmsg, const char *(*)(sqlite3 *))
#define sqlite3_bind_blob MN_API(MN_SLOT_bind_blob, int (*)(sqlite3_stmt *, int, const void *, int, void (*)(void *)))
#define sqlite3_bind_double MN_API(MN_SLOT_bind_double, int (*)(sqlite3_stmt *, int, double))
#define sqlite3_bind_int MN_API(MN_SLOT_bind_int, int (*)(sqlite3_stmt *, int, int))
#define sqlite3_bind_int64 MN_API(MN_SLOT_bind_int64, int (*)(sqlite3_stmt *, int, sqlite3_int64))
#define sqlite3_bind_null MN_API(MN_SLOT_bind_null, int (*)(sqlite3_stmt *, int))
#define sqlite3_bind_value MN_API(MN_SLOT_bind_value, int (*)(sqlite3_stmt *, int, const sqlite3_value *))
#define sqlite3_bind_text MN_API(MN_SLOT_bind_text, int (*)(sqlite3_stmt *, int, const char *, int, void (*)(void *)))
#define sqlite3_column_blob MN_API(MN_SLOT_column_blob, const void *(*)(sqlite3_stmt *, int))
#define sqlite3_column_bytes MN_API(MN_SLOT_column_bytes, int (*)(sqlite3_stmt *, int))
#define sqlite3_column_count MN_API(MN_SLOT_column_count, int (*)(sqlite3_stmt *))
#define sqlite3_column_double MN_API(MN_SLOT_column_double, double (*)(sqlite3_stmt *, int))
#define sqlite3_column_int MN_API(MN_SLOT_column_int, int (*)(sqlite3_stmt *, int))
#define sqlite3_column_int64 MN_API(MN_SLOT_column_int64, sqlite3_int64 (*)(sqlite3_stmt *, int))
#define sqlite3_column_text MN_API(MN_SLOT_column_text, const unsigned char *(*)(sqlite3_stmt *, int))
#define sqlite3_column_type MN_API(MN_SLOT_column_type, int (*)(sqlite3_stmt *, int))
#define sqlite3_value_blob MN_API(MN_SLOT_value_blob, const void *(*)(sqlite3_value *))
#define sqlite3_value_bytes MN_API(MN_SLOT_value_bytes, int (*)(sqlite3_value *))
#define sqlite3_value_double MN_API(MN_SLOT_value_double, double (*)(sqlite3_value *))
#define sqlite3_value_int MN_API(MN_SLOT_value_int, int (*)(sqlite3_value *))
#define sqlite3_value_int64 MN_API(MN_SLOT_value_int64, sqlite3_int64 (*)(sqlite3_value *))
#define sqlite3_value_text MN_API(MN_SLOT_value_text, const unsigned char *(*)(sqlite3_value *))
#define sqlite3_value_type MN_API(MN_SLOT_value_type, int (*)(sqlite3_value *))
#define sqlite3_result_blob MN_API(MN_SLOT_result_blob, void (*)(sqlite3_context *, const void *, int, void (*)(void *)))
#define sqlite3_result_double MN_API(MN_SLOT_result_double, void (*)(sqlite3_context *, double))
#define sqlite3_result_error MN_API(MN_SLOT_result_error, void (*)(sqlite3_context *, const char *, int))
#define sqlite3_result_int MN_API(MN_SLOT_result_int, void (*)(sqlite3_context *, int))
#define sqlite3_result_int64 MN_API(MN_SLOT_result_int64, void (*)(sqlite3_context *, sqlite3_int64))
#define sqlite3_result_null MN_API(MN_SLOT_result_null, void (*)(sqlite3_context *))
#define sqlite3_result_text MN_API(MN_SLOT_result_text, void (*)(sqlite3_context *, const char *, int, void (*)(void *)))
#define sqlite3_user_data MN_API(MN_SLOT_user_data, void *(*)(sqlite3_context *))
#define sqlite3_context_db_handle MN_API(MN_SLOT_context_db_handle, sqlite3 *(*)(sqlite3_context *))

#endif /* MN_SQLITE_ABI_H */
