/* Small file helpers (behaviour of reference src/utils.c:20-75). */
#include <ctype.h>
#include "utils.h"

FILE *openfile(const char *filename) {
    FILE *f = fopen(filename, "r");
    if (f == NULL) {
        fprintf(stderr, "error opening file %s\n", filename);
        perror("file could not be opened");
        exit(1);
    }
    return f;
}

unsigned int countlines(const char *filename) {
    FILE *f = openfile(filename);
    unsigned int n = 0;
    int c;
    while ((c = fgetc(f)) != EOF)
        if (c == '\n')
            n++;
    fclose(f);
    return n;
}

unsigned int get_column_count(const char *filename) {
    char line[10000];
    unsigned int count = 0;
    int in_token = 0, i;
    FILE *f = openfile(filename);
    if (fgets(line, sizeof line, f) == NULL) {
        fprintf(stderr, "error: file %s is empty!", filename);
        exit(1);
    }
    fclose(f);
    for (i = 0; line[i] != 0; i++) {
        if (isspace((unsigned char)line[i])) {
            in_token = 0;
        } else if (!in_token) {
            in_token = 1;
            count++;
        }
    }
    return count;
}
