/* tests/tools/splitcand_gate185.c -- DEV TOOL / TEST INFRASTRUCTURE (uses the CPU oracle's tables; never part of the product).
 * Host-side gate of the PAIR pool of the vienna-1.8.5 fill kernel (fold_lds_kernel.hip, splits_sparse185): counts the strict split candidates of dangles 1 and the
 * distinct pairs that realise them, and checks cell by cell that DML over the pooled pairs' four variants equals the dense minimum.
 *   gcc -O2 -o /tmp/gate185 tests/tools/splitcand_gate185.c -lm && /tmp/gate185 < windows.txt   (windows: tests/tools/dump_bench_windows.py) */
#include "../../oracle/lfold185.c"
/* d1: count distinct pairs (p,q) that realise a strict split candidate through one of their four dangle variants, and check the pair-pool identity:
   DML(i,j) = min(DML(i,j-1), min over pooled pairs (p,q) and variants (s,j') in {(p,q),(p-1,q),(p,q+1),(p-1,q+1)} with j'==j of fML(i,s-1) + variant value) */
int main(int argc, char **argv) {
    int span = 300; char line[4096]; double totc=0, totp=0, nw=0; long mxp=0, mism=0;
    while (fgets(line, sizeof line, stdin)) {
        int n = (int)strlen(line);
        while (n && (line[n-1]=='\n'||line[n-1]=='\r')) line[--n]=0;
        if (n < 10) continue;
        Fold F; F.n=n; F.M=span;
        F.seq=(char*)calloc(n+16,1); F.S=(int*)calloc(n+2,sizeof(int));
        for (int i=1;i<=n;i++){ char ch=(char)toupper((unsigned char)line[i-1]); if(ch=='T')ch='U'; F.seq[i]=ch; F.S[i]= ch=='A'?1:ch=='C'?2:ch=='G'?3:ch=='U'?4:0; }
        F.S[0]=F.S[n]; F.S[n+1]=F.S[1];
        size_t cells=(size_t)(n+2)*(span+2);
        F.c=(int*)malloc(cells*sizeof(int)); F.fML=(int*)malloc(cells*sizeof(int)); F.pt=(unsigned char*)calloc(cells,1); F.f3=(int*)calloc(n+span+8,sizeof(int));
        for(size_t x=0;x<cells;x++) F.c[x]=F.fML[x]=INF;
        for(int i=1;i<=n;i++) for(int j=i+TURN+1;j<=n&&j-i<=span-1;j++) F.pt[IDX(&F,i,j)]=(unsigned char)PAIR[F.S[i]][F.S[j]];
        fill(&F);
        const int *S = F.S;
        unsigned char *pooled=(unsigned char*)calloc(cells,1); int *dml=(int*)malloc(cells*sizeof(int));
        long ncand=0, npool=0;
        for(int i=1;i<=n;i++) for(int j=i+TURN+1;j<=n&&j<=i+span;j++){
            int v=mget(&F,i,j), d=DML(&F,i,j); dml[IDX(&F,i,j)]=d;
            int o=imin(imin(mget(&F,i+1,j),mget(&F,i,j-1)),d);
            if(v<o&&v<INF/2){ ncand++;
                /* which pair realises it: same order as the fill */
                int t, e, done=0;
                t=ptype(&F,i,j);     e = cget(&F,i,j)+MLintern(t); if(!done && t && e==v){ if(!pooled[IDX(&F,i,j)]){pooled[IDX(&F,i,j)]=1;npool++;} done=1; }
                t=ptype(&F,i+1,j);   e = cget(&F,i+1,j)+d5(t,S[i])+MLintern(t)+T99_ML_BASE; if(!done && t && e==v){ if(!pooled[IDX(&F,i+1,j)]){pooled[IDX(&F,i+1,j)]=1;npool++;} done=1; }
                t=ptype(&F,i,j-1);   e = cget(&F,i,j-1)+d3(t,S[j])+MLintern(t)+T99_ML_BASE; if(!done && t && e==v){ if(!pooled[IDX(&F,i,j-1)]){pooled[IDX(&F,i,j-1)]=1;npool++;} done=1; }
                t=ptype(&F,i+1,j-1); e = cget(&F,i+1,j-1)+d5(t,S[i])+d3(t,S[j])+MLintern(t)+2*T99_ML_BASE; if(!done && t && e==v){ if(!pooled[IDX(&F,i+1,j-1)]){pooled[IDX(&F,i+1,j-1)]=1;npool++;} done=1; }
                if(!done) mism+=1000000;
            }
        }
        /* identity with the pair pool */
        for (int d=TURN+1; d<=span && d<n; d++) for (int i=1;i<=n-d;i++){ int j=i+d; int best=(d-1>TURN)?dml[IDX(&F,i,j-1)]:INF;
            for (int dq=0; dq<=1; dq++) { int q=j-dq; /* pairs (p,q) with variant column j */
                for (int p=i+TURN+2; p<=q-TURN-1 && p<=n; p++) { if (q-p>span-1 || q-p<=TURN) continue; if(!pooled[IDX(&F,p,q)]) continue;
                    int t=ptype(&F,p,q); int val=cget(&F,p,q)+MLintern(t);
                    int e3 = dq? d3(t,S[q+1]) : 0;
                    /* s = p */
                    if (p-1-i >= TURN+1) best=imin(best, mget(&F,i,p-1)+val+e3);
                    /* s = p-1 */
                    if (p-2-i >= TURN+1 && p-1>=1) best=imin(best, mget(&F,i,p-2)+val+e3+d5(t,S[p-1]));
                } }
            int dense=dml[IDX(&F,i,j)]; if((dense<INF/2||best<INF/2)&&dense!=best) mism++; }
        totc+=ncand; totp+=npool; nw++; if(npool>mxp)mxp=npool;
        free(pooled); free(dml);
    }
    printf("windows %.0f candidates/window %.0f pooled pairs/window %.0f max %ld mismatches %ld\n", nw, totc/nw, totp/nw, mxp, mism);
    return 0;
}
