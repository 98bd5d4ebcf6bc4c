/* view_client.h -- the client half of `bgt view` through a resident host; static, so that the launcher (main.c) can use it
 * without loading libbgt.so and the HIP runtime behind it (13 ms of dynamic linking that a 4 ms query should not pay). */
#ifndef BGT_VIEW_CLIENT_H
#define BGT_VIEW_CLIENT_H
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <sys/socket.h>
#include <sys/un.h>

/* ------------------------------------------------------------------------------------------------
 * BGT_SERVER=<unix socket>: hand the query to a resident `bgt-server -u <socket>` instead of starting a HIP runtime and
 * building the images in this process (140-270 ms against the reference's 7 ms for a small query).
 * The request carries this process's stdout and stderr AS FILE DESCRIPTORS (SCM_RIGHTS), its working directory and
 * its arguments; the server runs the same view_run() on its resident images and writes straight into those descriptors,
 * so the bytes are those of a local run; the answer on the socket is the exit status.  No server there: run locally.
 * ------------------------------------------------------------------------------------------------ */
static int view_via_server(const char *path, int argc, char *argv[])
{
    struct sockaddr_un sa;
    struct msghdr mh;
    struct iovec iov;
    union { struct cmsghdr h; char buf[CMSG_SPACE(2 * sizeof(int))]; } cm;
    struct cmsghdr *c;
    char cwd[PATH_MAX], *req;
    size_t len = 0, cap, k;
    int fd, i, fds[2] = {1, 2};
    unsigned char status[2];
    ssize_t n;
    if (strlen(path) >= sizeof(sa.sun_path) || getcwd(cwd, sizeof(cwd)) == NULL) return -1;
    if ((fd = socket(AF_UNIX, SOCK_STREAM, 0)) < 0) return -1;
    memset(&sa, 0, sizeof(sa));
    sa.sun_family = AF_UNIX; strcpy(sa.sun_path, path);
    if (connect(fd, (struct sockaddr*)&sa, sizeof(sa)) < 0) { close(fd); return -1; }
    /* request: "BGTV1\0" cwd "\0" argc (decimal) "\0" argv[0] "\0" ... , preceded by its length (uint32) */
    cap = strlen(cwd) + 64;
    for (i = 0; i < argc; ++i) cap += strlen(argv[i]) + 1;
    req = (char*)malloc(cap + 4);
    len = 4;
    len += (size_t)sprintf(req + len, "BGTV1") + 1;
    len += (size_t)sprintf(req + len, "%s", cwd) + 1;
    len += (size_t)sprintf(req + len, "%d", argc) + 1;
    for (i = 0; i < argc; ++i) { k = strlen(argv[i]) + 1; memcpy(req + len, argv[i], k); len += k; }
    { const uint32_t body = (uint32_t)(len - 4); memcpy(req, &body, 4); }
    fflush(stdout); fflush(stderr);
    memset(&mh, 0, sizeof(mh)); memset(&cm, 0, sizeof(cm));
    iov.iov_base = req; iov.iov_len = len;
    mh.msg_iov = &iov; mh.msg_iovlen = 1;
    mh.msg_control = cm.buf; mh.msg_controllen = sizeof(cm.buf);
    c = CMSG_FIRSTHDR(&mh);
    c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(fds));
    memcpy(CMSG_DATA(c), fds, sizeof(fds));
    n = sendmsg(fd, &mh, MSG_NOSIGNAL);                             /* the descriptors travel with the first byte */
    if (n < 0) { free(req); close(fd); return -1; }
    for (k = (size_t)n; k < len; k += (size_t)n)
        if ((n = send(fd, req + k, len - k, MSG_NOSIGNAL)) <= 0) { free(req); close(fd); return 1; }
    free(req);
    for (k = 0; k < 2; k += (size_t)n) {                            /* {'S', status}: anything else is a lost server */
        do n = read(fd, status + k, 2 - k); while (n < 0 && errno == EINTR);
        if (n <= 0) break;
    }
    close(fd);
    if (k != 2 || status[0] != 'S') { fprintf(stderr, "[E::main_view] the server at '%s' went away before it answered.\n", path); return 1; }
    return status[1];
}

#define VIEW_FAIL(code) do { rc = (code); goto done; } while (0)
#endif
