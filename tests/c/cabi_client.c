/* A plain-C client of libsignerf_hip.so (r06): proves that include/signerf_hip.h is valid C99, that the library is usable without any
 * C++ / torch type at the boundary, and exercises the ABI handshake from the language a foreign binding would be written in.
 * Built and run by tests/test_cabi.py (gcc, dlopen; no GPU needed: every call made here returns before a device is touched).
 *   usage: cabi_client <path to libsignerf_hip.so>      prints "key value" lines, exit code 0 = every check passed */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "signerf_hip.h"

typedef int (*abi_fn)(void);
typedef int (*create_fn)(const SnFieldDesc*, SnHandle*);
typedef const char* (*err_fn)(SnHandle);
typedef size_t (*maskws_fn)(int32_t, int32_t);

static void nerfacto_desc(SnFieldDesc* d) {
    int i, l;
    memset(d, 0, sizeof(*d));
    d->struct_size = (uint32_t)sizeof(*d);
    d->main_field.num_levels = 16;
    d->main_field.features_per_level = 2;
    d->main_field.log2_hashmap_size = 19;
    d->main_field.hidden_dim = 64;
    d->main_field.num_layers = 2;
    d->main_field.out_dim = 16;
    for (l = 0; l < 16; ++l) d->main_field.scalings[l] = 16.0f + 8.0f * (float)l;
    d->geo_feat_dim = 15;
    d->hidden_dim_color = 64;
    d->appearance_embed_dim = 32;
    d->sh_levels = 4;
    d->num_proposals = 2;
    for (i = 0; i < 2; ++i) {
        d->proposals[i].num_levels = 5;
        d->proposals[i].features_per_level = 2;
        d->proposals[i].log2_hashmap_size = 17;
        d->proposals[i].hidden_dim = 16;
        d->proposals[i].num_layers = 2;
        d->proposals[i].out_dim = 1;
        for (l = 0; l < 5; ++l) d->proposals[i].scalings[l] = 16.0f + 20.0f * (float)l;
    }
    d->average_init_density = 0.01f;
    d->histogram_padding = 0.01f;
}

int main(int argc, char** argv) {
    void* lib;
    abi_fn abi;
    create_fn create;
    err_fn last_error;
    maskws_fn mask_ws;
    SnFieldDesc d;
    SnHandle h = 0;
    int st, bad = 0;
    if (argc < 2) return 2;
    lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        printf("dlopen_error %s\n", dlerror());
        return 3;
    }
    *(void**)(&abi) = dlsym(lib, "sn_abi_version");   /* (the POSIX idiom: ISO C has no object -> function pointer cast) */
    *(void**)(&create) = dlsym(lib, "sn_create");
    *(void**)(&last_error) = dlsym(lib, "sn_last_error");
    *(void**)(&mask_ws) = dlsym(lib, "sn_mask_workspace_bytes");
    if (!abi || !create || !last_error || !mask_ws) return 4;
    printf("abi_version %d header %d\n", abi(), SN_ABI_VERSION);
    bad |= abi() != SN_ABI_VERSION;
    printf("sizeof SnFieldDesc %zu SnRenderOpts %zu SnMaskOpts %zu SnDebugLayout %zu\n", sizeof(SnFieldDesc), sizeof(SnRenderOpts), sizeof(SnMaskOpts),
           sizeof(SnDebugLayout));
    printf("mask_workspace_64x64 %zu\n", mask_ws(64, 64));
    bad |= mask_ws(64, 64) == 0;
    /* 1. struct_size never set */
    nerfacto_desc(&d);
    d.struct_size = 0;
    st = create(&d, &h);
    printf("unset_size status %d text %s\n", st, last_error(0));
    bad |= st != SN_ERR_INVALID || !strstr(last_error(0), "was not set");
    /* 2. a caller newer than the library */
    nerfacto_desc(&d);
    d.struct_size += 64;
    st = create(&d, &h);
    printf("newer_caller status %d text %s\n", st, last_error(0));
    bad |= st != SN_ERR_INVALID || !strstr(last_error(0), "newer header");
    /* 3. an architecture the kernels are not written for */
    nerfacto_desc(&d);
    d.main_field.hidden_dim = 128;
    st = create(&d, &h);
    printf("bad_arch status %d text %s\n", st, last_error(0));
    bad |= st != SN_ERR_INVALID || !strstr(last_error(0), "hidden_dim");
    /* 4. a valid descriptor: SN_OK on a GPU box, SN_ERR_HIP ("no HIP device") where there is none -- never SN_ERR_INVALID */
    nerfacto_desc(&d);
    st = create(&d, &h);
    printf("valid status %d\n", st);
    bad |= !(st == SN_OK || st == SN_ERR_HIP);
    if (st == SN_OK) {
        typedef int (*destroy_fn)(SnHandle);
        destroy_fn destroy;
        *(void**)(&destroy) = dlsym(lib, "sn_destroy");
        if (destroy) destroy(h);
    }
    printf("result %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
