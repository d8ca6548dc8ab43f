/* _fastcall -- CPython binding of the two batch-1 calls of include/srbdqp.h, for the reference's control loop.
 *
 * `MPC.update(contact_horizon, c_horizon, p_com_horizon, x_current=, one_rollout=)` (g1_mujoco_sim/src/run_simulation.py:106) is one C call,
 * srbdqp_update_f64, on the library's pinned staging arrays.  Through ctypes + NumPy the Python side of that call cost 3.1 - 3.7 us of a ~55 us call
 * (two np.concatenate of the reference's per-step lists, five slice assignments, the ctypes trampoline, two result copies); here the lists are walked
 * with the C API and copied straight into the staging arrays, the library is entered through its function pointer, and the two results are fresh arrays
 * filled by memcpy: ~0.6 us.  Nothing in here computes: the module holds POINTERS the Python side got from the library (srbdqp_stage_ptrs) and the address
 * of srbdqp_update_f64 / srbdqp_solve_staged_f64; it does not link against libsrbdqp.so and has no fallback of its own -- anything it does not
 * recognise (other dtypes, shapes, nested lists) returns NotImplemented and the caller takes the general NumPy path of g1_locomotion_amd/mpc.py.
 *
 * Built by __graft_entry__.build() (gcc, in tree).  Optional: without it mpc.py binds the same C call through ctypes.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

typedef int (*update_fn)(void* h, const double* x0, const double* x_ref, const double* foot, const uint8_t* contact, const double* pcom,
                         double* u0_out, double* u_out, double* x_out, int32_t* status, int32_t* iters);
typedef int (*staged_fn)(void* h, int32_t B, int32_t use_pcom, int32_t use_warm, int32_t want_x, int32_t want_y);

typedef struct {
    update_fn update;
    staged_fn staged;
    void* handle;
    int N;
    double *x0, *xref, *foot, *pcom, *u, *x;     /* staging arrays of QP 0 (host addresses, srbdqp_stage) */
    uint8_t* contact;
    int32_t *status, *iters;
    double solve_time;                           /* seconds inside the last library call */
} Bound;

static void bound_free(PyObject* cap) { PyMem_Free(PyCapsule_GetPointer(cap, "srbdqp.bound")); }

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* bind(update_addr, staged_addr, handle_addr, N, x0, xref, foot, contact, pcom, u, x, status, iters) -> capsule */
static PyObject* fc_bind(PyObject* self, PyObject* args) {
    unsigned long long a[13];
    (void)self;
    if (!PyArg_ParseTuple(args, "KKKKKKKKKKKKK", &a[0], &a[1], &a[2], &a[3], &a[4], &a[5], &a[6], &a[7], &a[8], &a[9], &a[10], &a[11], &a[12])) return NULL;
    Bound* b = (Bound*)PyMem_Malloc(sizeof(Bound));
    if (!b) return PyErr_NoMemory();
    b->update = (update_fn)(uintptr_t)a[0]; b->staged = (staged_fn)(uintptr_t)a[1]; b->handle = (void*)(uintptr_t)a[2]; b->N = (int)a[3];
    b->x0 = (double*)(uintptr_t)a[4]; b->xref = (double*)(uintptr_t)a[5]; b->foot = (double*)(uintptr_t)a[6]; b->contact = (uint8_t*)(uintptr_t)a[7];
    b->pcom = (double*)(uintptr_t)a[8]; b->u = (double*)(uintptr_t)a[9]; b->x = (double*)(uintptr_t)a[10];
    b->status = (int32_t*)(uintptr_t)a[11]; b->iters = (int32_t*)(uintptr_t)a[12];
    b->solve_time = 0.0;
    if (!b->update || !b->handle || b->N <= 0 || !b->x0 || !b->xref || !b->foot || !b->contact || !b->pcom || !b->u || !b->x || !b->status || !b->iters) {
        PyMem_Free(b);
        PyErr_SetString(PyExc_ValueError, "_fastcall.bind: null address");
        return NULL;
    }
    return PyCapsule_New(b, "srbdqp.bound", bound_free);
}

/* n doubles of a float64 array of exactly n elements -> dst (a strided view, e.g. x_ref_hor[:, 3:6], through NumPy's own copy); 0 = not that */
static int take_f64(PyObject* o, double* dst, npy_intp n) {
    if (!PyArray_Check(o)) return 0;
    PyArrayObject* a = (PyArrayObject*)o;
    if (PyArray_TYPE(a) != NPY_FLOAT64 || PyArray_SIZE(a) != n) return 0;
    if (PyArray_IS_C_CONTIGUOUS(a)) { memcpy(dst, PyArray_DATA(a), (size_t)n * sizeof(double)); return 1; }
    PyArrayObject* c = (PyArrayObject*)PyArray_NewCopy(a, NPY_CORDER);
    if (!c) { PyErr_Clear(); return 0; }
    memcpy(dst, PyArray_DATA(c), (size_t)n * sizeof(double));
    Py_DECREF(c);
    return 1;
}

/* n contact flags (0 / non-zero) of a C-contiguous array of a plain numeric dtype -> dst */
static int take_flags(PyObject* o, uint8_t* dst, npy_intp n) {
    if (!PyArray_Check(o)) return 0;
    PyArrayObject* a = (PyArrayObject*)o;
    if (!PyArray_IS_C_CONTIGUOUS(a) || PyArray_SIZE(a) != n) return 0;
    const void* p = PyArray_DATA(a);
    npy_intp i;
    switch (PyArray_TYPE(a)) {
        case NPY_BOOL: case NPY_UINT8: case NPY_INT8: for (i = 0; i < n; ++i) dst[i] = ((const uint8_t*)p)[i] != 0; return 1;
        case NPY_INT64: case NPY_UINT64: for (i = 0; i < n; ++i) dst[i] = ((const int64_t*)p)[i] != 0; return 1;
        case NPY_INT32: case NPY_UINT32: for (i = 0; i < n; ++i) dst[i] = ((const int32_t*)p)[i] != 0; return 1;
        case NPY_FLOAT64: for (i = 0; i < n; ++i) dst[i] = ((const double*)p)[i] != 0.0; return 1;
        case NPY_FLOAT32: for (i = 0; i < n; ++i) dst[i] = ((const float*)p)[i] != 0.0f; return 1;
        default: return 0;
    }
}

/* a horizon given as ONE (N, per) array or as a list / tuple of N arrays of `per` entries (the reference's per-step lists, run_simulation.py:94-101) */
static int take_horizon_f64(PyObject* o, double* dst, int N, int per) {
    if (PyArray_Check(o)) return take_f64(o, dst, (npy_intp)N * per);
    if (!PyList_Check(o) && !PyTuple_Check(o)) return 0;
    if (PySequence_Fast_GET_SIZE(o) != N) return 0;
    PyObject** it = PySequence_Fast_ITEMS(o);
    for (int k = 0; k < N; ++k) if (!take_f64(it[k], dst + (size_t)k * per, per)) return 0;
    return 1;
}
static int take_horizon_flags(PyObject* o, uint8_t* dst, int N, int per) {
    if (PyArray_Check(o)) return take_flags(o, dst, (npy_intp)N * per);
    if (!PyList_Check(o) && !PyTuple_Check(o)) return 0;
    if (PySequence_Fast_GET_SIZE(o) != N) return 0;
    PyObject** it = PySequence_Fast_ITEMS(o);
    for (int k = 0; k < N; ++k) if (!take_flags(it[k], dst + (size_t)k * per, per)) return 0;
    return 1;
}

/* update(bound, contact_horizon, c_horizon, p_com_horizon, x_current, x_ref_hor, one_rollout) -> (u_opt0 (12, 1), x_opt1, status, rc) | NotImplemented */
static PyObject* fc_update(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 7) { PyErr_SetString(PyExc_TypeError, "_fastcall.update takes 7 arguments"); return NULL; }
    Bound* b = (Bound*)PyCapsule_GetPointer(args[0], "srbdqp.bound");
    if (!b) return NULL;
    const int N = b->N;
    /* the inputs go straight into the staging arrays (srbdqp_update_f64 leaves an argument that IS the staging array where it is) */
    if (!take_f64(args[4], b->x0, 13) || !take_f64(args[5], b->xref, (npy_intp)N * 13) || !take_horizon_f64(args[2], b->foot, N, 12) ||
        !take_horizon_flags(args[1], b->contact, N, 4))
        Py_RETURN_NOTIMPLEMENTED;
    const int use_pcom = args[3] != Py_None;
    if (use_pcom && !take_horizon_f64(args[3], b->pcom, N, 3)) Py_RETURN_NOTIMPLEMENTED;
    const int full = PyObject_IsTrue(args[6]);
    if (full < 0) return NULL;
    int rc;
    const double t0 = now_s();
    Py_BEGIN_ALLOW_THREADS
    rc = b->update(b->handle, b->x0, b->xref, b->foot, b->contact, use_pcom ? b->pcom : NULL, b->u, NULL, b->x, NULL, NULL);
    Py_END_ALLOW_THREADS
    b->solve_time = now_s() - t0;
    if (rc != 0) return Py_BuildValue("OOii", Py_None, Py_None, 0, rc);      /* the caller asks the library for the message */
    npy_intp du[2] = {12, 1}, dx[2] = {full ? N + 1 : 2, 13};
    PyObject* u0 = PyArray_SimpleNew(2, du, NPY_FLOAT64);
    PyObject* x1 = PyArray_SimpleNew(2, dx, NPY_FLOAT64);
    if (!u0 || !x1) { Py_XDECREF(u0); Py_XDECREF(x1); return NULL; }
    memcpy(PyArray_DATA((PyArrayObject*)u0), b->u, 12 * sizeof(double));
    memcpy(PyArray_DATA((PyArrayObject*)x1), b->x, (size_t)dx[0] * 13 * sizeof(double));
    PyObject* out = PyTuple_New(4);
    if (!out) { Py_DECREF(u0); Py_DECREF(x1); return NULL; }
    PyTuple_SET_ITEM(out, 0, u0);
    PyTuple_SET_ITEM(out, 1, x1);
    PyTuple_SET_ITEM(out, 2, PyLong_FromLong((long)b->status[0]));
    PyTuple_SET_ITEM(out, 3, PyLong_FromLong(0));
    return out;
}

/* solve_staged(bound, B, use_pcom, use_warm, want_x, want_y) -> rc : srbdqp_solve_staged_f64 through its function pointer */
static PyObject* fc_solve_staged(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 6) { PyErr_SetString(PyExc_TypeError, "_fastcall.solve_staged takes 6 arguments"); return NULL; }
    Bound* b = (Bound*)PyCapsule_GetPointer(args[0], "srbdqp.bound");
    if (!b) return NULL;
    if (!b->staged) { PyErr_SetString(PyExc_ValueError, "_fastcall.solve_staged: not bound"); return NULL; }
    long v[5];
    for (int i = 0; i < 5; ++i) { v[i] = PyLong_AsLong(args[1 + i]); if (v[i] == -1 && PyErr_Occurred()) return NULL; }
    int rc;
    const double t0 = now_s();
    Py_BEGIN_ALLOW_THREADS
    rc = b->staged(b->handle, (int32_t)v[0], (int32_t)v[1], (int32_t)v[2], (int32_t)v[3], (int32_t)v[4]);
    Py_END_ALLOW_THREADS
    b->solve_time = now_s() - t0;
    return PyLong_FromLong(rc);
}

static PyObject* fc_solve_time(PyObject* self, PyObject* cap) {
    (void)self;
    Bound* b = (Bound*)PyCapsule_GetPointer(cap, "srbdqp.bound");
    if (!b) return NULL;
    return PyFloat_FromDouble(b->solve_time);
}

static PyMethodDef methods[] = {
    {"bind", fc_bind, METH_VARARGS, "bind(update_addr, staged_addr, handle, N, x0, xref, foot, contact, pcom, u, x, status, iters) -> capsule"},
    {"update", (PyCFunction)(void (*)(void))fc_update, METH_FASTCALL, "update(bound, contact_horizon, c_horizon, p_com_horizon, x_current, x_ref_hor, one_rollout)"},
    {"solve_staged", (PyCFunction)(void (*)(void))fc_solve_staged, METH_FASTCALL, "solve_staged(bound, B, use_pcom, use_warm, want_x, want_y) -> rc"},
    {"solve_time", fc_solve_time, METH_O, "seconds inside the last library call of this binding"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastcall", "CPython binding of the batch-1 calls of libsrbdqp.so (pointers only; no compute)", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fastcall(void) {
    import_array();
    return PyModule_Create(&moddef);
}
