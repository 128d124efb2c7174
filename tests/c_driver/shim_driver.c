/* shim_driver.c -- TEST INFRASTRUCTURE.  Replays, from plain C, the exact call sequence of the Julia shim's single-`Mechanism`
 * drop-in (dojo.jl_amd/julia/DojoHIP.jl: enable! + hip_mehrotra!, i.e. the Dojo.mehrotra! override of SURVEY.md §7-3(c)):
 *
 *   dojo_create(topology, B = 1, fp64) -> dojo_set_options -> dojo_set_external_force(Fext, tau_ext)
 *   -> dojo_step_impulses(z, [JF2; Jtau2]) -> dojo_get_solution -> dojo_get_mu -> (write-back) -> dojo_destroy
 *
 * Julia cannot run in the build container, so this is how the boundary the shim binds is executed without Python: the
 * header is compiled as C, the structs are filled as C PODs, the library is dlopen'ed.  Input: a blob written by
 * tests/test_c_driver.py (counts, topology arrays, options, z, jf, fext); output: the raw results, which the test compares
 * with dojo_step(z, u) through the Python binding and with the oracle.
 *
 * usage: shim_driver <libdojo_hip.so> <in.bin> <out.bin>
 */
#include "dojo_hip.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LOAD(name) do { *(void**)(&p_##name) = dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 3; } } while (0)
#define RD(ptr, n) do { if (fread((ptr), 1, (n), f) != (size_t)(n)) { fprintf(stderr, "short read\n"); return 4; } } while (0)
#define CK(call) do { int rc_ = (call); if (rc_ != DOJO_OK) { fprintf(stderr, "%s -> %d: %s | handle: %s\n", #call, rc_, p_dojo_last_error(), h ? p_dojo_handle_error(h) : ""); return 5; } } while (0)

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s lib in out\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    const char* (*p_dojo_last_error)(void); const char* (*p_dojo_handle_error)(DojoHandle);
    int (*p_dojo_create)(const DojoTopology*, int32_t, int32_t, int32_t, DojoHandle*); void (*p_dojo_destroy)(DojoHandle);
    int (*p_dojo_set_options)(DojoHandle, const DojoSolverOptions*); int (*p_dojo_set_external_force)(DojoHandle, const void*);
    int (*p_dojo_step_impulses)(DojoHandle, const void*, const void*, void*, int32_t*, int32_t*);
    int (*p_dojo_get_solution)(DojoHandle, void*, void*, void*); int (*p_dojo_get_mu)(DojoHandle, double*); int (*p_dojo_get_dims)(DojoHandle, DojoDims*);
    LOAD(dojo_last_error); LOAD(dojo_handle_error); LOAD(dojo_create); LOAD(dojo_destroy); LOAD(dojo_set_options); LOAD(dojo_set_external_force);
    LOAD(dojo_step_impulses); LOAD(dojo_get_solution); LOAD(dojo_get_mu); LOAD(dojo_get_dims);

    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 4; }
    DojoTopology T; memset(&T, 0, sizeof T);
    int32_t cnt[4]; RD(cnt, sizeof cnt);
    T.n_bodies = cnt[0]; T.n_joints = cnt[1]; T.n_contacts = cnt[2];
    double hdr[5]; RD(hdr, sizeof hdr);
    T.timestep = hdr[0]; T.input_scaling = hdr[1]; T.gravity[0] = hdr[2]; T.gravity[1] = hdr[3]; T.gravity[2] = hdr[4];
    DojoBody* bodies = (DojoBody*)calloc(T.n_bodies > 0 ? T.n_bodies : 1, sizeof(DojoBody));
    DojoJoint* joints = (DojoJoint*)calloc(T.n_joints > 0 ? T.n_joints : 1, sizeof(DojoJoint));
    DojoContact* contacts = (DojoContact*)calloc(T.n_contacts > 0 ? T.n_contacts : 1, sizeof(DojoContact));
    RD(bodies, sizeof(DojoBody) * T.n_bodies); RD(joints, sizeof(DojoJoint) * T.n_joints); RD(contacts, sizeof(DojoContact) * T.n_contacts);
    T.bodies = bodies; T.joints = joints; T.contacts = contacts;
    DojoSolverOptions opts; RD(&opts, sizeof opts);
    const int nb = T.n_bodies;
    double* z = (double*)malloc(sizeof(double) * 13 * nb); double* jf = (double*)malloc(sizeof(double) * 6 * nb); double* fext = (double*)malloc(sizeof(double) * 6 * nb);
    RD(z, sizeof(double) * 13 * nb); RD(jf, sizeof(double) * 6 * nb); RD(fext, sizeof(double) * 6 * nb);
    fclose(f);

    DojoHandle h = NULL;
    CK(p_dojo_create(&T, 1, DOJO_DTYPE_F64, 0, &h));                      /* enable!(mechanism) */
    DojoDims D; CK(p_dojo_get_dims(h, &D));
    CK(p_dojo_set_options(h, &opts));                                     /* hip_mehrotra!: set_options! */
    CK(p_dojo_set_external_force(h, fext));                               /*                state.Fext / state.tau_ext */
    double* zn = (double*)malloc(sizeof(double) * 13 * nb); int32_t status = -1, iters = -1;
    CK(p_dojo_step_impulses(h, z, jf, zn, &status, &iters));              /*                the solve, controls as JF2 / Jtau2 */
    const int nji = D.n_joint_impulses > 0 ? D.n_joint_impulses : 1, ncs = D.n_contacts > 0 ? 8 * D.n_contacts : 1;
    double* vel = (double*)malloc(sizeof(double) * 6 * nb); double* ji = (double*)calloc(nji, sizeof(double)); double* cs = (double*)calloc(ncs, sizeof(double));
    CK(p_dojo_get_solution(h, vel, ji, cs));                              /*                write-back: vsol, omega_sol, impulses */
    double mu = -1.0;
    CK(p_dojo_get_mu(h, &mu));                                            /*                mechanism.mu, then set_entries! on the Julia side */
    /* a failing call reports on the handle it was made on */
    int bad = p_dojo_step_impulses(h, NULL, jf, zn, &status, &iters);
    int err_ok = bad != DOJO_OK && strstr(p_dojo_handle_error(h), "dojo_step_impulses") != NULL;
    p_dojo_destroy(h);

    FILE* g = fopen(argv[3], "wb");
    if (!g) { perror(argv[3]); return 4; }
    int32_t out_hdr[6] = {status, iters, D.n_joint_impulses, D.n_contacts, err_ok, (int32_t)sizeof(DojoJoint)};
    fwrite(out_hdr, sizeof out_hdr, 1, g); fwrite(&mu, sizeof mu, 1, g);
    fwrite(zn, sizeof(double), 13 * nb, g); fwrite(vel, sizeof(double), 6 * nb, g); fwrite(ji, sizeof(double), nji, g); fwrite(cs, sizeof(double), ncs, g);
    fclose(g);
    printf("shim_driver: status %d iters %d mu %.3e\n", status, iters, mu);
    return 0;
}
